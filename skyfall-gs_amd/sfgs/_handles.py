"""Shared rule of the storage-less tensor handles (sfgs.prepass.Deferred, sfgs.features.DeferredFeatures,
sfgs.viewdirs.LazyDirs, sfgs.sh.DeferredColor): which attribute READS the wrapper answers itself.

A handle stands for a tensor it has not computed (or, for the handle on `_xyz`, for a parameter). Its shape / dtype /
device / requires_grad / is_leaf are known from metadata and answered without materialising anything -- that is what lets
`pc.get_features.shape[0]` cost nothing. Every OTHER attribute (`.grad`, `.grad_fn`, `._version`, `.data`, `.T`, ...) is a
property of the tensor the handle stands for and is read from it (`LazyDirs(XYZ, p).grad` used to answer None
while `p.grad` was set -- the patched getter must behave like the reference's `return self._xyz`)."""

WRAPPER_PROPS = frozenset(("shape", "dtype", "device", "requires_grad", "is_leaf", "ndim", "layout", "is_cuda", "is_cpu",
                           "is_sparse", "is_sparse_csr", "is_quantized", "is_meta", "is_nested", "is_mkldnn", "is_xpu",
                           "is_mps", "is_xla", "is_ipu", "is_maia", "is_mtia", "is_vulkan", "is_ort", "names", "itemsize",
                           "nbytes"))


def property_name(func):
    """'grad' for torch.Tensor.grad.__get__ (the function __torch_function__ receives for an attribute read)."""
    return getattr(getattr(func, "__self__", None), "__name__", None)


def answered_by_wrapper(func):
    return property_name(func) in WRAPPER_PROPS
