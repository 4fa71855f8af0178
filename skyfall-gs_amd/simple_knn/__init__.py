"""simple_knn -- drop-in package; the reference imports `from simple_knn._C import distCUDA2`
(scene/gaussian_model.py:25)."""
