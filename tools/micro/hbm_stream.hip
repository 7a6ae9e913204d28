// Micro-benchmark: attainable HBM streaming rates on one MI355X for the access mixes of the hot
// path -- read-only, write-only, 1R+1W, 2R+1W (row kernels' shape), 4R+3W (k_pcg_update_dir) --
// 16-byte accesses, grid-stride, over 512 MiB arrays (well beyond the 256 MiB Infinity Cache).
// Build: hipcc -O3 --offload-arch=gfx950 hbm_stream.hip -o hbm_stream ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int NR, int NW>
__global__ void __launch_bounds__(1024) k(const v4* const* __restrict__ in, v4* const* __restrict__ out,
                                          size_t n, float a) {
    const v4* ip[NR > 0 ? NR : 1];
    v4* op[NW > 0 ? NW : 1];
    for (int j = 0; j < NR; ++j) ip[j] = in[j];
    for (int j = 0; j < NW; ++j) op[j] = out[j];
    v4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v4 s = {a, a, a, a};
        for (int j = 0; j < NR; ++j) s += ip[j][i] * (a + j);
        if (NW == 0) acc += s;
        for (int j = 0; j < NW; ++j) op[j][i] = s + (float)j;
    }
    if (NW == 0 && acc.x == 1234.5f) ((v4*)ip[0])[0] = acc;      // never true: keeps the loads alive
}

template <int NR, int NW> void run(const char* name, size_t n, v4** bufs, v4** dptr) {
    std::vector<v4*> h(NR + NW);
    for (int j = 0; j < NR + NW; ++j) h[j] = bufs[j];
    hipMemcpy(dptr, h.data(), sizeof(v4*) * (NR + NW), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0; int bg = 0, bb = 0;
    for (int block : {64, 128, 256, 512, 1024})
        for (int per_cu : {1, 2, 4, 8, 16, 32}) {
            const int grid = 256 * per_cu;
            hipLaunchKernelGGL((k<NR, NW>), dim3(grid), dim3(block), 0, 0, (const v4* const*)dptr, dptr + NR, n, 0.5f);
            hipEventRecord(e0);
            for (int it = 0; it < 5; ++it)
                hipLaunchKernelGGL((k<NR, NW>), dim3(grid), dim3(block), 0, 0, (const v4* const*)dptr, dptr + NR, n, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double tbs = (double)(NR + NW) * n * 16 / (ms * 1e-3) / 1e12;
            if (tbs > best) { best = tbs; bg = grid; bb = block; }
            printf("    %dR+%dW block %4d grid %5d: %.2f TB/s\n", NR, NW, block, grid, tbs);
        }
    printf("%-12s %dR+%dW: best %.2f TB/s (grid %d x %d threads)\n", name, NR, NW, best, bg, bb);
}

int main() {
    const size_t n = (size_t)512 << 20 >> 4;          // 512 MiB per array, in 16-byte elements
    v4* bufs[7];
    for (auto& b : bufs) { hipMalloc(&b, n * 16); hipMemset(b, 0, n * 16); }
    v4** dptr; hipMalloc(&dptr, sizeof(v4*) * 8);
    run<1, 0>("read", n, bufs, dptr);
    run<0, 1>("write", n, bufs, dptr);
    run<1, 1>("copy", n, bufs, dptr);
    run<2, 1>("2R+1W", n, bufs, dptr);
    run<3, 1>("3R+1W", n, bufs, dptr);
    run<4, 3>("update_dir", n, bufs, dptr);
    return 0;
}
