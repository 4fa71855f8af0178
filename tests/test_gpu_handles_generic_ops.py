"""tests/test_handles_generic_ops.py ON THE GPU (VERDICT r4 "weak" 2): the same table of operations applied to the storage-less
handles and to the tensors they stand for, with everything on `cuda` -- `__torch_function__` / `__torch_dispatch__` of a wrapper
subclass that reports a cuda device, materialisation through device kernels, and the backward running on the autograd
engine's device worker thread instead of the calling thread."""
import pytest
import torch

import test_handles_generic_ops as cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handles():
    return {**cpu._features("cuda"), **cpu._dirs("cuda")}


@pytest.mark.parametrize("op", sorted(cpu.OPS))
@pytest.mark.parametrize("kind", cpu.KINDS)
def test_any_operation_on_a_cuda_handle_equals_the_operation_on_the_tensor(handles, kind, op):
    make_handle, make_real, leaves = handles[kind]
    assert make_handle().is_cuda and all(t.is_cuda for t in leaves)
    cpu.check_operation(handles, kind, op)


def test_backward_through_a_cuda_handle_from_another_host_thread(handles):
    """render() may be driven from a worker thread (a data-loader style trainer): the recorded view-direction statements are
    evaluated, and differentiated, from a thread that is not the one that created the handle."""
    import threading
    make_handle, make_real, leaves = handles["dirs"]
    out = {}

    def work():
        for t in leaves:
            t.grad = None
        (make_handle() * 2.0).sum().backward()
        out["g"] = [t.grad.clone() for t in leaves]
    th = threading.Thread(target=work)
    th.start()
    th.join()
    for t in leaves:
        t.grad = None
    (make_real() * 2.0).sum().backward()
    for a, t in zip(out["g"], leaves):
        assert torch.equal(a, t.grad)
