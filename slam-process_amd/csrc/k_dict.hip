// Correlation of mean-RSS matrices with two beam dictionaries, C[f] = Phi_RX^T . M[f] . Phi_TX, batched over frames.
//
// Replaces the first line of every round of LoSEstimator.estimate_los_path (heatmap_gemini_v4.py:207-209:
// `correlation = self.Phi_RX.T @ res_mat @ self.Phi_TX`) and of NLoSEstimator.estimate_nlos_paths (:304), the one dense
// contraction in the reference: Phi_RX [n_ue x ga] and Phi_TX [n_bs x gd] are Gaussian beam dictionaries over search
// grids of a tenth of a degree (ga, gd about 890 for the 64-beam table, construct_dictionary :182-198, :283-296), M is
// the [n_ue x n_bs] mean-RSS matrix or the residual of a round.  The rest of those estimators (arg-max, NNLS, masks,
// plots) stays on the host and is not part of this library.
//
// f64 throughout, on the matrix cores: v_mfma_f64_16x16x4_f64 (A: lane l holds A[l & 15][l >> 4], B: B[l >> 4][l & 15],
// C/D: column l & 15, row (l >> 4) + 4 * register).  A workgroup of four waves owns 64 rows of the search grid:
//   step 1   T = Phi_RX^T[64 rows] . M               (64 x n_ue) . (n_ue x n_bs), operands straight from memory (a
//            fifteenth of the work), T through LDS into the layout of an A operand;
//   step 2   C[64 rows, :] = T . Phi_TX              every wave keeps its 16 rows of T as A fragments in registers
//            (16 k-steps) and walks the columns 32 at a time; Phi_TX's 32-column slices alternate between two LDS areas
//            (where T was), the next one fetched into registers before this one's products are issued.
// 34 KB of LDS and at most 128 registers: four workgroups per CU, four waves per SIMD.
// Beams beyond n_ue / n_bs and grid points beyond ga / gd are zeros in the operands, so any n_ue, n_bs <= 64 and any
// ga, gd are served.  Per frame at 64 beams and 886 x 886 grid points: 0.108 GFLOP and 6.3 MB of output, i.e. about as
// long on the f64 matrix rate (78.6 TFLOP/s) as on the HBM write stream.

#include "pipeline.hpp"

namespace {

constexpr int DC_K = 64;               // beams per side (padded)
constexpr int DC_LD = DC_K + 1;        // row stride of T in doubles: 16 rows of one column land in 16 different bank pairs
constexpr int DC_NC = 32;              // output columns per slice
constexpr int DC_LS = DC_NC + 1;       // row stride of a slice
constexpr int DC_SLICE = DC_K * DC_LS; // doubles per slice area
static_assert(2 * DC_SLICE >= DC_K * DC_LD, "T fits where the two slices go");
typedef double dc_acc __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
void k_dict_corr(const double *phi_rx, int n_ue, int ga, const double *phi_tx, int n_bs, int gd, const double *m, double *out)
{
    __shared__ double buf[2 * DC_SLICE];     // step 1: T (64 grid rows x 64 beams); step 2: two slices of Phi_TX
    const int f = blockIdx.y, g0 = blockIdx.x * 64, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const double *mf = m + (size_t)f * n_ue * n_bs;
    double *of = out + (size_t)f * ga * gd;

    // step 1: this wave's 16 grid rows of T
    {
        const int g = g0 + 16 * wv + lr;
        dc_acc acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 4
        for (int ks = 0; ks < DC_K / 4; ks++) {
            const int u = 4 * ks + lq;
            const double a = (g < ga && u < n_ue) ? phi_rx[(size_t)u * ga + g] : 0.0;       // A[row g][k u] = Phi_RX[u][g]
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int b = 16 * nt + lr;
                const double mv = (u < n_ue && b < n_bs) ? mf[(size_t)u * n_bs + b] : 0.0;  // B[k u][col b] = M[u][b]
                acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, mv, acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) buf[(16 * wv + lq + 4 * r) * DC_LD + 16 * nt + lr] = acc[nt][r];
    }
    __syncthreads();
    // this wave's rows of T as A fragments: a[ks] = T[row lr][k 4 ks + lq]
    double a[DC_K / 4];
#pragma unroll
    for (int ks = 0; ks < DC_K / 4; ks++) a[ks] = buf[(16 * wv + lr) * DC_LD + 4 * ks + lq];
    // step 2
    double nx[DC_K * DC_NC / 256];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int q = 0; q < DC_K * DC_NC / 256; q++) {
            const int i = tid + 256 * q, v = i / DC_NC, j = i % DC_NC;
            nx[q] = (v < n_bs && c0 + j < gd) ? phi_tx[(size_t)v * gd + c0 + j] : 0.0;
        }
    };
    auto stash = [&](double *area) {
#pragma unroll
        for (int q = 0; q < DC_K * DC_NC / 256; q++) {
            const int i = tid + 256 * q;
            area[(i / DC_NC) * DC_LS + (i % DC_NC)] = nx[q];
        }
    };
    fetch(0);
    __syncthreads();                                       // every wave has its fragments: T may be overwritten
    stash(buf);
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < gd; c0 += DC_NC, cur ^= 1) {
        const double *bs_ = buf + cur * DC_SLICE;
        const bool more = c0 + DC_NC < gd;
        if (more) fetch(c0 + DC_NC);
        dc_acc acc[DC_NC / 16] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int ks = 0; ks < DC_K / 4; ks++) {
#pragma unroll
            for (int nt = 0; nt < DC_NC / 16; nt++)
                acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], bs_[(4 * ks + lq) * DC_LS + 16 * nt + lr], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < DC_NC / 16; nt++) {
            const int col = c0 + 16 * nt + lr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = g0 + 16 * wv + lq + 4 * r;
                if (row < ga && col < gd) of[(size_t)row * gd + col] = acc[nt][r];
            }
        }
        if (more) stash(buf + (cur ^ 1) * DC_SLICE);       // the area the slice before this one was read from
        __syncthreads();
    }
}

}  // namespace

int dict_correlate(slam_ctx *ctx, const double *phi_rx, int n_ue, int ga, const double *phi_tx, int n_bs, int gd,
                   const double *m, uint32_t n_frames, double *out)
{
    if (n_frames == 0) return SLAM_OK;
    SLAM_LAUNCH(ctx, "dict_corr", k_dict_corr, dim3((ga + 63) / 64, n_frames), dim3(256), phi_rx, n_ue, ga, phi_tx, n_bs, gd,
                m, out);
    return SLAM_OK;
}
