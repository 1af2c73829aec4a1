"""Drop-in for the `simple_knn` package (reference import: scene_reconstruction/gaussian_mesh.py:26)."""
