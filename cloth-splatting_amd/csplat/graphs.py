"""A step of the batched rasterizer recorded into hipGraphs and replayed -- ONE implementation for the path `bench.py` times, the path
`tests/test_raster_gpu.py::test_config2_full_size_faith_replay_vs_oracle` holds to the oracle, and the capacities `csplat.train.CapturedStep`
records with ("the timed path is not the tested path").

What replaces a host read.  The reference's rasterizer call reads `num_rendered` back between its two phases
(/root/reference/gaussian_renderer/__init__.py:156-164 -> upstream `rasterize_points.cu`); a stream capture cannot contain a host read.
`csplat_forward_views_faith` launches both phases with caller-given capacities and leaves a device word saying whether the counts fitted;
every later kernel of the step leaves an unfitting view alone.  The caller checks the word whenever it synchronises anyway."""
import contextlib as _contextlib

import torch

from . import native as _n

MARGIN = 8          # capacities = counts + counts / MARGIN (+ a constant)
LIST_CAP = 8192     # longest tile list the in-LDS tile sort takes (csplat_raster.hip)


@_contextlib.contextmanager
def capture(graph):
    """`torch.cuda.graph(graph)` with Python's cyclic collector held off for the duration of the capture.  A collection that starts in the
    middle of a capture (allocation counts crossing the collector's threshold) frees whatever unreachable tensors it finds -- memory of
    ANOTHER recording's private pool, events of the previous eager step -- and freeing those is an operation the capturing stream refuses:
    the process aborts inside the collector (seen in round 5 when a second step shape was recorded while the first shape's graph was
    alive).  Garbage is collected before the capture starts, the collector resumes after it ends."""
    import gc
    was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield graph
    finally:
        if was_on:
            gc.enable()


def caps_from_counts(counts, margin=MARGIN):
    """per-view (tile instances R, longest tile list L, non-empty tiles B) -> the capacities a launch on faith is sized with"""
    R = max(int(c[0]) for c in counts)
    L = max(int(c[1]) for c in counts)
    B = max(int(c[2]) for c in counts)
    return (R + R // margin + 4096, min(L + L // 4 + 64, LIST_CAP), B + B // margin + 16)


def counts_of_eager(fn):
    """runs fn() (a step that goes through `rasterize_views` eagerly) and returns (fn's result, the per-view [R, L, B] counts its batched
    forward left on the device)"""
    import diff_gaussian_rasterization as dgr
    with dgr.forward_mode(keep_info=True):
        out = fn()
    if not dgr.LAST_INFO:
        raise RuntimeError("counts_of_eager: the step did not go through the batched forward (rasterize_views)")
    return out, torch.stack(list(dgr.LAST_INFO)).cpu().tolist()


class ReplayedSteps:
    """`fn()` -- one step of the workload: batched forward (`rasterize_views`), loss, backward -- recorded into G hipGraphs that are
    replayed round-robin.  Same launches, same work as the eager fn(); what is gone is the host: ~35 launches and one read of the counts
    per step.  G graphs, not one: every recording has its own output buffers, so G steps are in flight before a result is overwritten.

        rs = ReplayedSteps(fn, device)      # one eager fn() for the counts the capacities come from
        rs.record()
        rs.step() ...                       # replays
        rs.check()                          # raises unless every replay so far did its work (device validity words, scratch epoch)
    """

    def __init__(self, fn, device, G=4, margin=MARGIN):
        self.fn, self.dev, self.G = fn, device, int(G)
        _out, self.counts = counts_of_eager(fn)
        self.caps = caps_from_counts(self.counts, margin)
        self.graphs, self.valid, self.info, self.outs = [], [], [], []
        self.k = 0
        self.epoch = None

    def record(self):
        import diff_gaussian_rasterization as dgr
        torch.cuda.synchronize(self.dev)
        self.graphs, self.valid, self.info, self.outs = [], [], [], []
        for _ in range(self.G):
            valid = torch.zeros(1, dtype=torch.int32, device=self.dev)
            faith = {"caps": self.caps, "valid": valid}
            g = torch.cuda.CUDAGraph()
            with dgr.forward_mode(faith=faith, replay_device=self.dev):
                with capture(g):
                    out = self.fn()
            self.graphs.append(g); self.valid.append(valid); self.info.append(faith["info"]); self.outs.append(out)
        self.epoch = _n.SCRATCH_EPOCH[0]
        self.k = 0
        return self

    def step(self):
        """one replay; returns what fn() returned when THAT graph was recorded (static tensors the replay has refilled)"""
        if self.epoch != _n.SCRATCH_EPOCH[0]:
            raise RuntimeError("ReplayedSteps: a scratch cache was evicted since the recording (csplat.native.evict_scratch): the graphs "
                               "point into freed memory -- record() again")
        i = self.k % len(self.graphs)
        self.graphs[i].replay()
        self.k += 1
        return self.outs[i]

    def last(self):
        """index of the graph the most recent step() replayed"""
        return (self.k - 1) % len(self.graphs)

    def all_valid(self):
        return all(int(v.item()) == 1 for v in self.valid[:max(1, min(self.k, len(self.valid)))])

    def replay_counts(self, i=None):
        """the per-view [R, L, B] counts graph i's last replay left on the device"""
        i = self.last() if i is None else i
        return torch.stack(list(self.info[i])).cpu().tolist()

    def check(self):
        torch.cuda.synchronize(self.dev)
        if not self.all_valid():
            raise RuntimeError("ReplayedSteps: a replayed step reported counts beyond its capacities (nothing was computed)")
        if self.epoch != _n.SCRATCH_EPOCH[0]:
            raise RuntimeError("ReplayedSteps: scratch evicted under live recordings")
