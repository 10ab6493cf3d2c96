// LDS / register canary: does a NEIGHBOUR process's kernel on the same compute unit change this workgroup's LDS or registers?
// Every workgroup fills `lds_kb` KiB of LDS with a pattern that encodes (word index), keeps 8 VGPRs of pattern per lane, and re-reads
// both for `spin` rounds; any word that changed is reported (workgroup, word index, value found, round) and repaired.  Run it beside
// other processes (tools/lab/race_matrix.sh canary_*): a clean solo run + reports beside a neighbour = the neighbour writes LDS it
// does not own.  usage: lds_canary [seconds = 20] [lds_kb = 48] [workgroups = 512]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Report { unsigned wg, word, found, round, kind, cu; };
constexpr int MAXREP = 4096;

__host__ __device__ __forceinline__ unsigned pat(unsigned i) { return 0xC0DE0000u ^ (i * 2654435761u) ^ (i >> 3); }

__global__ __launch_bounds__(256) void k_canary(Report *rep, int *nrep, int words, int spin)
{
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < words; i += 256) lds[i] = pat(i);
    unsigned r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { r[j] = pat(0x100000u + tid * 8 + j); asm volatile("" : "+v"(r[j])); }
    __syncthreads();
    unsigned cu;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
    for (int it = 0; it < spin; it++) {
        for (int i = tid; i < words; i += 256) {
            const unsigned v = lds[i];
            if (v != pat(i)) {
                const int s = atomicAdd(nrep, 1);
                if (s < MAXREP) rep[s] = Report{blockIdx.x, (unsigned)i, v, (unsigned)it, 0u, cu};
                lds[i] = pat(i);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            asm volatile("" : "+v"(r[j]));
            if (r[j] != pat(0x100000u + tid * 8 + j)) {
                const int s = atomicAdd(nrep, 1);
                if (s < MAXREP) rep[s] = Report{blockIdx.x, (unsigned)(tid * 8 + j), r[j], (unsigned)it, 1u, cu};
                r[j] = pat(0x100000u + tid * 8 + j);
            }
        }
        __builtin_amdgcn_s_sleep(8);
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
    const int lds_kb = argc > 2 ? atoi(argv[2]) : 48;
    const int wgs = argc > 3 ? atoi(argv[3]) : 512;
    const int words = lds_kb * 256;
    Report *rep; int *nrep;
    CK(hipMalloc(&rep, sizeof(Report) * MAXREP)); CK(hipMalloc(&nrep, 4)); CK(hipMemset(nrep, 0, 4));
    CK(hipFuncSetAttribute((const void *)k_canary, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    int total = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipLaunchKernelGGL(k_canary, dim3(wgs), dim3(256), lds_kb * 1024, 0, rep, nrep, words, 200);
        CK(hipDeviceSynchronize());
        launches++;
        int n; CK(hipMemcpy(&n, nrep, 4, hipMemcpyDeviceToHost));
        if (n) {
            std::vector<Report> h(n < MAXREP ? n : MAXREP);
            CK(hipMemcpy(h.data(), rep, sizeof(Report) * h.size(), hipMemcpyDeviceToHost));
            printf("canary launch %ld: %d changed words\n", launches, n);
            for (size_t i = 0; i < h.size() && i < 48; i++)
                printf("  %s wg %u word %u (byte %u) found %08x want %08x round %u hw_id %08x\n", h[i].kind ? "VGPR" : "LDS ", h[i].wg, h[i].word, h[i].word * 4,
                       h[i].found, h[i].kind ? pat(0x100000u + h[i].word) : pat(h[i].word), h[i].round, h[i].cu);
            total += n;
            CK(hipMemset(nrep, 0, 4));
        }
    }
    printf("canary: %ld launches of %d workgroups x %d KiB LDS in %.1f s: %d changed words\n", launches, wgs, lds_kb, seconds, total);
    return 0;
}
