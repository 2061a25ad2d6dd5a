"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of ATen's CPU fp32 `torch.sum` order.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product path (transformer-quantization_amd/) never does.

Why it exists: the reference's MSE range estimator hands `torch.sum(torch.sum(err.view(len(data), -1),
dim=1))` (an fp32 value, /root/reference/quantization/range_estimators.py:248-256) to
`scipy.optimize.minimize_scalar` (:296-327, :422-470) and to a numpy argmin (:370, :405); the thresholds
it returns therefore depend on the exact fp32 summation order of a THIRD-PARTY dependency of the
reference: PyTorch's CPU sum kernel (`aten/src/ATen/native/cpu/SumKernel.cpp`, `cascade_sum` ->
`vectorized_inner_sum` / `scalar_inner_sum` -> `row_sum` -> `multi_row_sum`; torch 2.10.0 installed in the
build container, the reference pins torch 1.4.0 in README.md:42-45 which predates the cascade kernel).
That source is not under /root/reference, so its published algorithm is restated here and PINNED against
the live `torch.sum` of the build container by tests/test_oracle_golden.py::test_aten_sum_restatement
(this module is "the oracle of the oracle": `oracle/tq_oracle.py` itself simply calls torch.sum).

Algorithm for one contiguous fp32 row of n elements (Vectorized<float>::size() == 8 in the sum kernel of
this build even when torch reports AVX512 capability, ilp_factor == 4, num_levels == 4):
  n < 8 : 4 scalar accumulators over groups of 4 elements (one group at most), left-over elements added to
          accumulator 0, then accumulators 1..3 added to accumulator 0.
  n >= 8: the row is n // 8 vectors; 4 vector accumulators (32 columns) take vectors 4 i + k in "steps"
          i = 0 .. n // 32 - 1 through a 4-level cascade (`multi_row_sum`): level 0 takes
          L = 2 ** max(4, ceil_log2(steps) // 4) steps, then is added into level 1 and cleared; level j-1
          is added into level j whenever the step index is a multiple of L ** j; at the end
          ((l0 + l1) + l2) + l3; the n // 8 % 4 left-over vectors are added to vector accumulator 0; vector
          accumulators 1..3 are added to accumulator 0; a scalar accumulator adds the n % 8 tail elements
          and then the 8 lanes of accumulator 0, in that order.
Reductions whose OUTPUT has a single element and whose input has >= 32768 elements are split over threads
by TensorIterator (`parallel_reduce` two-pass) and depend on the thread count; the reference's loss never
hits that case for BERT / MobileBERT tensors (len(data) <= 30522 row sums).
"""
import numpy as np

F32 = np.float32
VEC = 8        # lanes of the vector type the sum kernel is compiled for
ILP = 4
GRAIN = 32768  # at::internal::GRAIN_SIZE


def ceil_log2(x):
    """c10::utils::CeilLog2"""
    if x <= 2:
        return 1
    return int(x - 1).bit_length()


def level_power(steps):
    return max(4, ceil_log2(steps) // 4)


def multi_row_sum(a):
    """a: fp32 [size, ncols] -> [ncols]; column-wise cascade sum (SumKernel.cpp multi_row_sum)."""
    size, nc = a.shape
    p = level_power(size)
    step = 1 << p
    mask = step - 1
    acc = np.zeros((4, nc), dtype=F32)
    i = 0
    while i + step <= size:
        for _ in range(step):
            acc[0] = acc[0] + a[i]
            i += 1
        for j in range(1, 4):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = 0
            if (i & (mask << (j * p))) != 0:
                break
    while i < size:
        acc[0] = acc[0] + a[i]
        i += 1
    for j in range(1, 4):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def row_sum(row):
    """fp32 sum of a contiguous 1-D fp32 array in ATen's order (single thread)."""
    row = np.ascontiguousarray(row, dtype=F32)
    n = row.shape[0]
    if n < VEC:                                        # scalar_inner_sum
        size_ilp = n // ILP
        part = multi_row_sum(row[:size_ilp * ILP].reshape(size_ilp, ILP)).copy()
        for i in range(size_ilp * ILP, n):
            part[0] = F32(part[0] + row[i])
        for k in range(1, ILP):
            part[0] = F32(part[0] + part[k])
        return F32(part[0])
    vec = n // VEC                                      # vectorized_inner_sum
    vecs = row[:vec * VEC].reshape(vec, VEC)
    steps = vec // ILP
    part = multi_row_sum(vecs[:steps * ILP].reshape(steps, ILP * VEC)).reshape(ILP, VEC).copy()
    for i in range(steps * ILP, vec):
        part[0] = part[0] + vecs[i]
    for k in range(1, ILP):
        part[0] = part[0] + part[k]
    fin = F32(0)
    for k in range(vec * VEC, n):
        fin = F32(fin + row[k])
    for k in range(VEC):
        fin = F32(fin + part[0][k])
    return fin


def sum_rows(a2d):
    """torch.sum(a2d, dim=1) for a contiguous fp32 [rows, n] array."""
    a2d = np.ascontiguousarray(a2d, dtype=F32)
    return np.array([row_sum(r) for r in a2d], dtype=F32)


def loss_sum(err, per_channel_loss=False):
    """The two sums of MSE_Estimator.loss_fx (range_estimators.py:250-256) on err = (data - y) ** 2."""
    err = np.ascontiguousarray(err, dtype=F32)
    rows = err.shape[0] if err.ndim else 1
    t = sum_rows(err.reshape(rows, -1))
    return t if per_channel_loss else row_sum(t)
