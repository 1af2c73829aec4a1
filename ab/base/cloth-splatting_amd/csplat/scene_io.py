"""Blender / D-NeRF style scene loader: `transforms_{train,test}.json` + images -> the camera records the
reference's Scene builds its Cameras from.  Restates /root/reference/scene_reconstruction/dataset_readers.py:
  read_timeline                 :386-401
  read_cameras_from_transforms  :268-385   (readCamerasFromTransforms)
  camera_from_info              scene_reconstruction/cameras.py:57-68 through csplat.synthetic.camera_matrices
Quirks reproduced, not fixed (the golden fixture tests/golden/scene_io.npz holds what the reference returns):
  * CameraInfo.width / .height are image.shape[1] / image.shape[2] of the [3,H,W] tensor, i.e. H and W swapped (:382);
  * the alpha-composited image is quantised by TRUNCATION (`np.array(arr * 255.0, dtype=np.byte)`, :371), not rounding;
  * frames whose file name is not r_<view>_<time> get their ids from the sorted unique transform matrices / times (:331-333);
  * `mapper` (read_timeline) is accepted and unused: the raw `time` of the frame is kept (:347-348);
  * the optic-flow side-car is never read upstream (`if False`, :273) and is not read here.
Host-side I/O: no GPU work, Pillow for the PNG decode."""
import json
import os
from pathlib import Path
from types import SimpleNamespace
from typing import NamedTuple, Optional

import numpy as np
import torch


class CameraInfo(NamedTuple):
    uid: int
    R: np.ndarray
    T: np.ndarray
    FovY: float
    FovX: float
    image: torch.Tensor
    image_path: str
    image_name: str
    width: int
    height: int
    time: float
    view_id: int
    time_id: int
    flow: Optional[np.ndarray] = None
    mask: Optional[torch.Tensor] = None


def read_timeline(path):
    times = []
    for split in ("train", "test"):
        with open(os.path.join(path, f"transforms_{split}.json")) as f:
            times += [frame["time"] for frame in json.load(f)["frames"]]
    times = sorted(set(times))
    top = max(times)
    return {t: t / top for t in times}, top


def _pil_to_torch(img):
    """utils/general_utils.py:21-30 without the resize branch"""
    t = torch.from_numpy(np.array(img)) / 255.0
    return t.permute(2, 0, 1) if t.dim() == 3 else t.unsqueeze(-1).permute(2, 0, 1)


def read_cameras_from_transforms(path, transformsfile, white_background, extension=".png", mapper=None, time_skip=None,
                                 view_skip=None):
    from PIL import Image
    mask_dir = os.path.join(path, "masks_gripper")
    mask_dir = mask_dir if os.path.exists(mask_dir) else None
    with open(os.path.join(path, transformsfile)) as f:
        contents = json.load(f)
    fovx, fovy = contents["camera_angle_x"], contents["camera_angle_y"]
    frames = contents["frames"]
    unique_times = np.unique([frame["time"] for frame in frames])
    unique_transforms = np.unique(np.stack([np.array(frame["transform_matrix"]) for frame in frames]), axis=0)
    kept_times = unique_times[::time_skip] if time_skip is not None else None
    out = []
    for idx, frame in enumerate(frames):
        if kept_times is not None and frame["time"] not in kept_times:
            continue
        file_path = frame["file_path"]
        known = (".png", ".jpg", ".jpeg")
        if not file_path.endswith(known):
            file_path += extension
        file_name = file_path.split("/")[-1]
        if file_path.endswith(known):
            file_name = file_name.split(".")[0]
        parts = file_name.split("_")
        if len(parts) > 2:
            view_id, time_id = int(parts[-2]), int(parts[-1])
        else:
            view_id = np.where(np.all(unique_transforms == np.array(frame["transform_matrix"]), axis=1))[0][0]
            time_id = np.where(unique_times == frame["time"])[0][0]
        if view_skip is not None and view_id % view_skip != 0:
            continue
        # NeRF / Blender camera-to-world (Y up, Z back) -> COLMAP axes (Y down, Z forward), then world-to-camera
        c2w = np.array(frame["transform_matrix"])
        c2w[:3, 1:3] *= -1
        w2c = np.linalg.inv(c2w)
        R = np.transpose(w2c[:3, :3])      # stored transposed ('glm' convention of the CUDA code)
        T = w2c[:3, 3]
        image_path = os.path.join(path, os.path.join(path, file_path))
        image_name = Path(image_path).stem
        rgba = np.array(Image.open(image_path).convert("RGBA")) / 255.0
        bg = np.array([1, 1, 1]) if white_background else np.array([0, 0, 0])
        arr = rgba[:, :, :3] * rgba[:, :, 3:4] + bg * (1 - rgba[:, :, 3:4])
        image = _pil_to_torch(Image.fromarray((arr * 255.0).astype(np.uint8)))       # truncation, as upstream's int8 cast
        mask = None
        if mask_dir:
            mask = 1.0 - _pil_to_torch(Image.open(os.path.join(mask_dir, image_name + ".png")))
        out.append(CameraInfo(uid=idx, R=R, T=T, FovY=fovy, FovX=fovx, image=image, image_path=image_path, image_name=image_name,
                              width=image.shape[1], height=image.shape[2], time=frame["time"], view_id=int(view_id),
                              time_id=int(time_id), flow=None, mask=mask))
    return out


def read_blender_scene(path, white_background, extension=".png", time_skip=None, view_skip=None):
    """the camera part of readNerfSyntheticInfo (:402-416): (train, test, video or None, timestamp mapper, max time)"""
    mapper, max_time = read_timeline(path)
    train = read_cameras_from_transforms(path, "transforms_train.json", white_background, extension, mapper, time_skip, view_skip)
    test = read_cameras_from_transforms(path, "transforms_test.json", white_background, extension, mapper, time_skip, view_skip)
    video = None
    if os.path.exists(os.path.join(path, "video.json")):
        video = read_cameras_from_transforms(path, "video.json", white_background, extension, mapper, 1, 1)
    return train, test, video, mapper, max_time


def camera_from_info(info, device="cuda"):
    """what gaussian_renderer.render() reads from a Camera (scene_reconstruction/cameras.py:17-68): matrices in the reference's
    row-vector convention, image size taken from the image tensor ([3, H, W])."""
    from .synthetic import camera_matrices
    wv, full, center = camera_matrices(np.asarray(info.R), np.asarray(info.T), info.FovX, info.FovY)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=device)  # noqa: E731
    H, W = int(info.image.shape[1]), int(info.image.shape[2])
    return SimpleNamespace(uid=info.uid, image_height=H, image_width=W, FoVx=info.FovX, FoVy=info.FovY,
                           world_view_transform=t(wv), full_proj_transform=t(full), camera_center=t(center), time=float(info.time),
                           original_image=info.image.clamp(0.0, 1.0).to(device=device, dtype=torch.float32),
                           mask=None if info.mask is None else info.mask.to(device=device, dtype=torch.float32),
                           view_id=info.view_id, time_id=info.time_id, image_name=info.image_name)
