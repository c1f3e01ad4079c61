// tools/ubench5.hip -- what ONE iteration's memory phases of the workgroup solve kernel (config C4: k_ldp_wg<4>) can get out of the memory
// system: 256 persistent workgroups of 8 waves, each re-reading ITS OWN working set every "iteration" with 16-byte loads, 16 in flight per
// lane (the kernel's scan depth): a 480 KB fp32 image of M (the screening scan) and `rows` active rows of 1.6 KB twice (primal step, Gram
// column).  Two footprints: every workgroup keeps ONE working set (256 x ~0.64 MB = 164 MB: within the 256 MB Infinity Cache, as the
// kernel's image + row cache mostly are while a problem iterates) or rotates over SETS working sets (beyond it: HBM).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench5.hip -o tools/ubench5.bin ; tools/ubench5.bin [workgroups] [iterations] [waves per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kImage = 480 * 1024, kRow = 1600, kRows = 100, kDepth = 16;



template <int DEPTH>
__device__ __forceinline__ float stream(const float4 *base, int count, int tid, int T)
{
    float acc = 0.f;
    for (int i0 = tid; i0 < count; i0 += T * DEPTH) {
        float4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) { const int i = i0 + u * T; v[u] = base[i < count ? i : i0]; }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    return acc;
}

__global__ __launch_bounds__(512) void k_iter(const float4 *mem, size_t set_f4, int sets, int iters, float *out)
{
    const int tid = threadIdx.x, T = blockDim.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float4 *ws = mem + ((size_t)blockIdx.x * sets + (it % sets)) * set_f4;
        acc += stream<kDepth>(ws, kImage / 16, tid, T);                         // the screening scan: the whole image
        __syncthreads();
        const float4 *rows = ws + kImage / 16;
        acc += stream<kDepth>(rows, kRows * kRow / 16, tid, T);                 // primal step: every active row
        __syncthreads();
        acc += stream<kDepth>(rows, kRows * kRow / 16, tid, T);                 // Gram column: every active row again
        __syncthreads();
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 200;
    const int waves = argc > 3 ? atoi(argv[3]) : 8;      // (round 6: 512 workgroups x 4 waves = the residency of the tiered launch, two workgroups per CU)
    const size_t set_bytes = (size_t)kImage + (size_t)kRows * kRow, set_f4 = set_bytes / 16;
    const double per_iter = (double)kImage + 2.0 * kRows * kRow;
    float *out;
    hipMalloc(&out, 4);
    for (int sets : {1, 4}) {
        float4 *mem;
        const size_t bytes = (size_t)wgs * sets * set_bytes;
        if (hipMalloc(&mem, bytes) != hipSuccess) { printf("no memory for %zu bytes\n", bytes); return 1; }
        hipMemset(mem, 0, bytes);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_iter, dim3(wgs), dim3(64 * waves), 0, 0, mem, set_f4, sets, 20, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_iter, dim3(wgs), dim3(64 * waves), 0, 0, mem, set_f4, sets, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double total = per_iter * iters * wgs;
        printf("%d workgroups x %d waves, %d working set(s) each (%.0f MB in all), %d iterations of %.0f KB: %.2f ms -> %.2f TB/s, %.1f B/clk/CU at 2.4 GHz, %.0f cycles per iteration\n",
               wgs, waves, sets, bytes / 1e6, iters, per_iter / 1024, ms, total / ms / 1e9, total / wgs / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / iters);
        hipFree(mem);
    }
    return 0;
}
