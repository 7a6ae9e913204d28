// Micro-benchmark: HBM bandwidth of strided "piece" accesses (development aid).
// A 128 MiB array viewed as [NV][NX] complex64 (NX = 4096 contiguous).  A block handles
// G consecutive i (piece = G*8 bytes) for all v: piece (v, i0) at offset (v*NX + i0)*8.
// Lanes cover a piece with 16-byte accesses (G/2 lanes per piece).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int G, bool WRITE>
__global__ void __launch_bounds__(1024) k_piece(float4* buf, int NV, int NX, float4* sink, int LD = 0) {
    if (LD) {      // padded leading dimension (elements of 8 bytes): piece (v, i0) at (v*LD + i0)*8 -- partition camping test
        constexpr int LP = G / 2;
        const int i0 = blockIdx.x * G;
        const int lp = threadIdx.x % LP, vi = threadIdx.x / LP;
        const int vstep = blockDim.x / LP;
        float4 acc = make_float4(0, 0, 0, 0);
        for (int v = vi; v < NV; v += vstep) {
            float4* p = buf + ((size_t)v * LD + i0) / 2 + lp;
            if (WRITE) *p = make_float4(v, lp, i0, 1.f);
            else { float4 t = *p; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
        }
        if (!WRITE && acc.x == 12345.f) sink[0] = acc;
        return;
    }
    constexpr int LP = G / 2;              // lanes per piece (16 B each)
    const int i0 = blockIdx.x * G;
    const int lp = threadIdx.x % LP, vi = threadIdx.x / LP;
    const int vstep = blockDim.x / LP;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int v = vi; v < NV; v += vstep) {
        float4* p = buf + ((size_t)v * NX + i0) / 2 + lp;
        if (WRITE) *p = make_float4(v, lp, i0, 1.f);
        else { float4 t = *p; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
    }
    if (!WRITE && acc.x == 12345.f) sink[0] = acc;
}

template <int G, bool WRITE>
float run_ld(float4* buf, int NV, int NX, int LD, float4* sink, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t band = (size_t)NV * LD / 2;
    for (int c = 0; c < 8; ++c) k_piece<G, WRITE><<<NX / G, 1024>>>(buf + c * band, NV, NX, sink, LD);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        for (int c = 0; c < 8; ++c) k_piece<G, WRITE><<<NX / G, 1024>>>(buf + c * band, NV, NX, sink, LD);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / (reps * 8);
}

template <int G, bool WRITE>
float run(float4* buf, int NV, int NX, float4* sink, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // 8 different 128 MiB "bands" per repetition: 1 GiB streamed, far beyond the 256 MiB MALL
    const size_t band = (size_t)NV * NX / 2;
    for (int c = 0; c < 8; ++c) k_piece<G, WRITE><<<NX / G, 1024>>>(buf + c * band, NV, NX, sink);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        for (int c = 0; c < 8; ++c) k_piece<G, WRITE><<<NX / G, 1024>>>(buf + c * band, NV, NX, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / (reps * 8);
}

int main() {
    const int NV = 4097, NX = 4096;
    const size_t bytes = (size_t)NV * NX * 8;
    float4 *buf, *sink;
    CK(hipMalloc(&buf, bytes * 8 + (size_t)8 * NV * 1024 * 8));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, bytes * 8));
    const int reps = 5;
    float ms;
#define RUN(G) \
    ms = run<G, true>(buf, NV, NX, sink, reps); \
    printf("piece %4d B  write: %7.3f ms  %7.1f GB/s", G * 8, ms, bytes / ms / 1e6); \
    ms = run<G, false>(buf, NV, NX, sink, reps); \
    printf("   read: %7.3f ms  %7.1f GB/s\n", ms, bytes / ms / 1e6);
    RUN(2) RUN(4) RUN(8) RUN(16) RUN(32) RUN(64) RUN(128)
    // partition camping: the same 64- and 128-byte pieces with the line stride padded by 0 / 16 / 32 / 64 / 272 elements
    for (int pad : {0, 16, 32, 64, 272}) {
        const int LD = NX + pad;
        float w8 = run_ld<8, true>(buf, NV, NX, LD, sink, reps), r8 = run_ld<8, false>(buf, NV, NX, LD, sink, reps);
        float w16 = run_ld<16, true>(buf, NV, NX, LD, sink, reps), r16 = run_ld<16, false>(buf, NV, NX, LD, sink, reps);
        printf("stride %6d B: 64-B pieces write %7.1f read %7.1f GB/s   128-B pieces write %7.1f read %7.1f GB/s\n",
               LD * 8, bytes / w8 / 1e6, bytes / r8 / 1e6, bytes / w16 / 1e6, bytes / r16 / 1e6);
    }
    return 0;
}
