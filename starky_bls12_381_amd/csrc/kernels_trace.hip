// On-device trace expansion: the host generator's write log (trace_log.h) -> the column-major
// matrix values[col][row] the prover starts from.  Replaces the 4.8 GB host fill + H2D copy + transpose of FinalExp
// by a ~150 MB upload and this kernel.
//
// A record is (col, first_row, run, n, v[n]): the limb vector v sits at columns col..col+n-1 on rows
// first_row..first_row+run-1.  Records never disagree on a cell, so they are expanded in parallel without ordering.
// One 64-lane wave per group of records: short records (run * n <= 64 cells, the per-row products and carries) take
// one lane per cell; long runs (replicated inputs) are walked row-fastest so that a wave writes whole 512-byte runs
// of a column.  The matrix is zeroed first (hipMemsetAsync in the caller).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace starkhip {

static constexpr unsigned TRACE_RECORDS_PER_WAVE = 8;

__global__ __launch_bounds__(64) void expand_trace_kernel(const uint32_t* __restrict__ words, const uint32_t* __restrict__ offsets,
                                                          size_t n_records, gl_t* __restrict__ values, size_t n_rows) { STARKHIP_PRIO_ENTRY
    const size_t first = (size_t)blockIdx.x * TRACE_RECORDS_PER_WAVE;
    const unsigned lane = threadIdx.x;
    for (unsigned k = 0; k < TRACE_RECORDS_PER_WAVE; k++) {
        const size_t r = first + k;
        if (r >= n_records) return;  // wave-uniform
        const uint32_t* rec = words + offsets[r];
        const uint32_t col = rec[0], row0 = rec[1], run = rec[2], n = rec[3];
        if (run == 0 || n == 0) continue;  // wave-uniform: a one-row run whose only cell the filler cleared again (TraceLog::set takes the row back)
        // lane -> (limb, row) ONCE per record: a run shorter than the wave packs 64 / run limbs side by side (rows fastest, so a limb's
        // run is one contiguous piece of its column) and steps that many limbs per store; a long run walks the rows of one limb after
        // the other.  (The first version divided by the run length for every cell: 25 of its 40 instructions per store.)
        if (run < 64) {
            const uint32_t per = 64 / run;                 // limbs per store (wave-uniform)
            const uint32_t l0 = lane / run, row = row0 + lane - l0 * run;
            if (l0 < per) {
                gl_t* dst = values + (size_t)(col + l0) * n_rows + row;
                for (uint32_t limb = l0; limb < n; limb += per, dst += (size_t)per * n_rows) *dst = rec[4 + limb];
            }
        } else {
            for (uint32_t limb = 0; limb < n; limb++) {
                const gl_t v = rec[4 + limb];
                gl_t* dst = values + (size_t)(col + limb) * n_rows + row0;
                for (uint32_t r2 = lane; r2 < run; r2 += 64) dst[r2] = v;
            }
        }
    }
}

// cells the generator cleared after writing them (trace_log.h, TraceLog::set): applied after the expansion, in stream order
__global__ void zero_cells_kernel(const uint32_t* __restrict__ col_row, size_t n_cells, gl_t* __restrict__ values, size_t n_rows) { STARKHIP_PRIO_ENTRY
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n_cells) values[(size_t)col_row[2 * i] * n_rows + col_row[2 * i + 1]] = 0;
}

hipError_t launch_zero_cells(const uint32_t* col_row, size_t n_cells, gl_t* values, size_t n_rows, hipStream_t st) {
    if (n_cells == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_cells_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, st, col_row, n_cells, values, n_rows);
    return hipGetLastError();
}

hipError_t launch_expand_trace(const uint32_t* words, const uint32_t* offsets, size_t n_records, gl_t* values, size_t n_rows, hipStream_t st) {
    if (n_records == 0) return hipSuccess;
    const size_t blocks = (n_records + TRACE_RECORDS_PER_WAVE - 1) / TRACE_RECORDS_PER_WAVE;
    hipLaunchKernelGGL(expand_trace_kernel, dim3((unsigned)blocks), dim3(64), 0, st, words, offsets, n_records, values, n_rows);
    return hipGetLastError();
}

}  // namespace starkhip
