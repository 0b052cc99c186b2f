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
        const uint32_t cells = run * n;  // <= 8192 * 24
        // cell j: row = row0 + j % run, limb = j / run: consecutive lanes -> consecutive rows of one column
        for (uint32_t j = lane; j < cells; j += 64) {
            const uint32_t limb = j / run, row = row0 + j % run;
            values[(size_t)(col + limb) * n_rows + row] = rec[4 + limb];
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
