// pce_align.hip -- dynamic time warping of token x frame cost matrices (R8 alignment step).
//
// Replaces the DTW that turns Whisper cross-attention into word timestamps
// (Code/Aligners/use_whisper_timestamped.py:163 -> whisper_timestamped / openai-whisper timing.py,
// third-party): cost[i+1][j+1] = x[i][j] + min(cost[i][j], cost[i][j+1], cost[i+1][j]) with the
// trace preference diagonal, up, left exactly as openai-whisper's dtw_cpu writes it
// ("c0 < c1 and c0 < c2 -> 0; elif c1 < c0 and c1 < c2 -> 1; else 2"), boundary trace[0,:] = 2,
// trace[:,0] = 1, back-tracking from (N, M).  fp64 adds and compares only, so the path indices are
// bit-identical to the CPU recurrence on the same matrix.
//
// One workgroup per matrix, thread = row, the anti-diagonals i + j = const sweep the matrix; the two
// previous diagonals live in LDS (double buffered), the 1-byte trace goes to global memory and is walked
// back by one lane.  N <= 1024 rows (Whisper: <= 448 tokens), any number of columns.
#include "pce_internal.h"

namespace {

constexpr int DTW_MAXN = 1024;

// matrix b: N = n_rows[b] (or n_rows_all), M = n_cols[b], element (i, j) at x[b * x_stride + i * ld + j];
// trace / path buffers are strided by the batch maxima (N_max, M_max).
__global__ __launch_bounds__(DTW_MAXN) void k_dtw(const double *__restrict__ x, int64_t x_stride, int ld, const int *__restrict__ n_rows,
                                                 const int *__restrict__ n_cols, int N_max, int M_max, unsigned char *__restrict__ trace,
                                                 int *__restrict__ path_i, int *__restrict__ path_j, int *__restrict__ path_len)
{
    __shared__ double diag[3][DTW_MAXN + 1];      // cost on diagonals d-2, d-1, d (index = row of the padded matrix)
    const int b = blockIdx.x, t = threadIdx.x;
    const int N = n_rows ? n_rows[b] : N_max, M = n_cols ? n_cols[b] : M_max;
    if (N <= 0 || M <= 0) { if (t == 0) path_len[b] = 0; return; }
    const double *xb = x + (size_t)b * (size_t)x_stride;
    unsigned char *tr = trace + (size_t)b * (size_t)(N_max + 1) * (size_t)(M_max + 1);
    const double INF = __builtin_huge_val();
    // padded cost matrix C[(N+1) x (M+1)]: C[0][0] = 0, rest of row 0 / column 0 = inf.
    // diagonal D (of the padded matrix) holds C[r][D - r]; thread t owns padded row r = t + 1.
    for (int r = t; r <= N; r += blockDim.x) { diag[0][r] = INF; diag[1][r] = INF; diag[2][r] = INF; }
    __syncthreads();
    if (t == 0) diag[0][0] = 0.0;                 // diagonal 0: C[0][0]
    // diagonal 1: C[0][1] = inf, C[1][0] = inf (already inf)
    __syncthreads();
    const int r = t + 1;
    for (int D = 2; D <= N + M; D++) {
        double *cur = diag[D % 3]; const double *p1 = diag[(D - 1) % 3], *p2 = diag[(D - 2) % 3];
        const int c = D - r;                      // padded column
        if (t < N && c >= 1 && c <= M) {
            const double c0 = p2[r - 1], c1 = p1[r - 1], c2 = p1[r];      // C[r-1][c-1], C[r-1][c], C[r][c-1]
            double cm; unsigned char tt;
            if (c0 < c1 && c0 < c2) { cm = c0; tt = 0; }
            else if (c1 < c0 && c1 < c2) { cm = c1; tt = 1; }
            else { cm = c2; tt = 2; }
            cur[r] = xb[(size_t)(r - 1) * ld + (c - 1)] + cm;
            tr[(size_t)r * (M + 1) + c] = tt;
        } else if (t < N) {
            cur[r] = INF;
        }
        if (t == 0) cur[0] = INF;                 // C[0][D] = inf for D >= 1
        __syncthreads();
    }
    // back-trace (one lane): trace[0][:] = 2, trace[:][0] = 1
    __threadfence_block();
    if (t == 0) {
        int i = N, j = M, n = 0;
        int *pi = path_i + (size_t)b * (N_max + M_max), *pj = path_j + (size_t)b * (N_max + M_max);
        while (i > 0 || j > 0) {
            pi[n] = i - 1; pj[n] = j - 1; n++;
            const int tt = (i == 0) ? 2 : (j == 0) ? 1 : tr[(size_t)i * (M + 1) + j];
            if (tt == 0) { i--; j--; } else if (tt == 1) { i--; } else { j--; }
        }
        // reverse in place
        for (int a = 0, z = n - 1; a < z; a++, z--) {
            const int ti = pi[a], tj = pj[a]; pi[a] = pi[z]; pj[a] = pj[z]; pi[z] = ti; pj[z] = tj;
        }
        path_len[b] = n;
    }
}

} // namespace

// device-resident batch (used by the Whisper alignment path): all pointers are device pointers
int pce_dtw_launch(pce_ctx *c, const double *d_x, int64_t x_stride, int ld, const int *d_rows, const int *d_cols, int N_max, int M_max,
                   int batch, unsigned char *d_trace, int *d_pi, int *d_pj, int *d_pl)
{
    if (N_max > DTW_MAXN) return pce_fail(c, PCE_E_LIMIT, "DTW supports at most %d rows", DTW_MAXN);
    KernelTimer t(c, PCE_K_DTW);
    const int threads = ((N_max + 63) / 64) * 64;
    hipLaunchKernelGGL(k_dtw, dim3((unsigned)batch), dim3((unsigned)threads), 0, c->stream, d_x, x_stride, ld, d_rows, d_cols, N_max, M_max,
                       d_trace, d_pi, d_pj, d_pl);
    PCE_HIP(c, hipGetLastError());
    return PCE_OK;
}

extern "C" {

int pce_dtw(pce_ctx *c, const double *x, int32_t n_rows, int32_t n_cols, int32_t batch, int32_t *path_i, int32_t *path_j, int32_t *path_len)
{
    if (!c || !x || !path_i || !path_j || !path_len || n_rows <= 0 || n_cols <= 0 || batch <= 0) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    const size_t cells = (size_t)batch * n_rows * n_cols, pl = (size_t)batch * (size_t)(n_rows + n_cols);
    DevBuf dx, dtr, dpi, dpj, dpl;
    PCE_HIP(c, dx.reserve(sizeof(double) * cells));
    PCE_HIP(c, dtr.reserve((size_t)batch * (size_t)(n_rows + 1) * (size_t)(n_cols + 1)));
    PCE_HIP(c, dpi.reserve(sizeof(int) * pl)); PCE_HIP(c, dpj.reserve(sizeof(int) * pl)); PCE_HIP(c, dpl.reserve(sizeof(int) * (size_t)batch));
    PCE_HIP(c, hipMemcpyAsync(dx.p, x, sizeof(double) * cells, hipMemcpyHostToDevice, c->stream));
    int rc = pce_dtw_launch(c, dx.as<double>(), (int64_t)n_rows * n_cols, n_cols, nullptr, nullptr, n_rows, n_cols, batch,
                            dtr.as<unsigned char>(), dpi.as<int>(), dpj.as<int>(), dpl.as<int>());
    if (rc) return rc;
    PCE_HIP(c, hipMemcpyAsync(path_i, dpi.p, sizeof(int) * pl, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipMemcpyAsync(path_j, dpj.p, sizeof(int) * pl, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipMemcpyAsync(path_len, dpl.p, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    dx.release(); dtr.release(); dpi.release(); dpj.release(); dpl.release();
    return PCE_OK;
}

} // extern "C"
