// Stable sort of (cell, value) pairs by cell for the real-valued pivot (k_render.hip, pivot_kahan).
//
// pandas' pivot_table(aggfunc=mean) (compute_rssi_matrix, excel_heatmap_v3.py:297-299) adds a cell's values in row
// order, so the rows of a cell have to stay in their order while they are brought together.  A least-significant-digit
// radix sort keeps equal keys in input order; this is rocPRIM's, used as a library routine on a side path (tables of
// RSS values that are not whole numbers, through the host-pointer call only).  It lives in a file of its own so the
// template instantiation is compiled once and the kernels of the hot path do not wait for it.
#include <cstring>

#include "pipeline.hpp"

#include <rocprim/device/device_radix_sort.hpp>

size_t sort_cells_temp_bytes(uint32_t n, unsigned bits)
{
    size_t bytes = 0;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                   (const double *)nullptr, (double *)nullptr, n, 0u, bits, (hipStream_t)0);
    return e == hipSuccess ? bytes : 0;
}

int sort_cells(slam_ctx *ctx, void *tmp, size_t tmp_bytes, const uint32_t *key_in, uint32_t *key_out, const double *val_in,
               double *val_out, uint32_t n, unsigned bits)
{
    const hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, n, 0u, bits, ctx->stream);
    if (e != hipSuccess) return slam_fail(ctx, SLAM_E_HIP, "stable sort by cell failed: %s", hipGetErrorString(e));
    return SLAM_OK;
}
