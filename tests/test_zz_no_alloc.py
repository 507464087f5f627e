"""spmv_hip_norm2_dev_noalloc -- the entry point that allocates nothing -- against
spmv_hip_norm2_dev, through the C ABI.  Both run the same two kernels over the same number of
partial sums, so the results must agree bit for bit: fp64 and fp32, two calls in a row with the
second on a smaller input (the built-in scratch array keeps the first call's partial sums), an empty
input, and the refusal of a size beyond the stated maximum.
"""
import numpy as np
import pytest
import torch

import spmv_amd as S

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _both(n, v, dtype):
    L = S.lib()
    out = torch.full((2,), float("nan"), dtype=torch.float64, device=DEV)
    assert L.spmv_hip_norm2_dev(n, v.data_ptr(), dtype, out[0:].data_ptr(), 0) == 0, L.spmv_hip_last_error()
    assert L.spmv_hip_norm2_dev_noalloc(n, v.data_ptr(), dtype, out[1:].data_ptr(), 0) == 0, L.spmv_hip_last_error()
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("dtype,td", [(S.F64, torch.float64), (S.F32, torch.float32)])
def test_noalloc_norm_equals_the_allocating_one_on_two_calls_in_a_row(orc, dtype, td):
    assert S.lib().spmv_hip_init(0) == 0
    g = torch.Generator(device=DEV).manual_seed(21)
    big = torch.rand(5_000_011, dtype=td, device=DEV, generator=g) - 0.5          # all 1024 partial sums in use
    small = torch.rand(70_001, dtype=td, device=DEV, generator=g) * 3.0           # 274 of them: the rest is stale
    tiny = torch.rand(5, dtype=td, device=DEV, generator=g)
    for v in (big, small, tiny, big):
        a, b = _both(v.numel(), v, dtype)
        assert a.tobytes() == b.tobytes()
        want = float(torch.linalg.vector_norm(v.to(torch.float64)))
        assert abs(b - want) <= 1e-12 * want
        # the driver's vec_norm (driver_spmv.c:21-26) as the oracle restates it: a sequential sum, so by tolerance
        host = np.ascontiguousarray(v.to(torch.float64).cpu().numpy())
        ref = orc.lib().orc_vec_norm(len(host), host)
        assert abs(b - ref) <= 1e-12 * ref
    a, b = _both(0, tiny, dtype)                                                   # empty input: norm 0
    assert a == 0.0 and b == 0.0


def test_noalloc_norm_runs_on_a_side_stream_and_refuses_what_it_cannot_hold():
    L = S.lib()
    v = torch.arange(1, 1001, dtype=torch.float64, device=DEV)
    out = torch.zeros(1, dtype=torch.float64, device=DEV)
    st = torch.cuda.Stream(device=DEV)
    st.wait_stream(torch.cuda.current_stream())
    assert L.spmv_hip_norm2_dev_noalloc(1000, v.data_ptr(), S.F64, out.data_ptr(), st.cuda_stream) == 0
    st.synchronize()
    assert float(out) == float(np.sqrt(np.sum(np.arange(1, 1001, dtype=np.float64) ** 2)))
    assert L.spmv_hip_norm2_dev_noalloc(2 ** 31, v.data_ptr(), S.F64, out.data_ptr(), 0) != 0
    assert b"outside 0..2147483647" in L.spmv_hip_last_error()
    assert L.spmv_hip_norm2_dev_noalloc(-1, v.data_ptr(), S.F64, out.data_ptr(), 0) != 0
    assert L.spmv_hip_norm2_dev_noalloc(10, v.data_ptr(), 7, out.data_ptr(), 0) != 0
    assert L.spmv_hip_norm2_dev_noalloc(10, None, S.F64, out.data_ptr(), 0) != 0
