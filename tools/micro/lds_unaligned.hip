// Micro-benchmark: ds_read_b128 at 16-byte-aligned vs 4-byte-aligned LDS addresses (gfx950).
//   hipcc -O3 --offload-arch=gfx950 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
// Question behind it: could the first layer keep its patch WITHOUT the 3x kw-slot duplication and read its
// A fragments as overlapping 16-byte windows at dword offsets, as the HBM side already does?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void lds_read_kernel(int iters, int misalign, int stride, uint32_t* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* w = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < 16384; i += 256) w[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t addr = (uint32_t)(lane * stride + misalign * 4) + (threadIdx.x >> 6) * 1024;     // byte address
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    {   // which words does one read return?  (lane 0 of wave 0 reads byte address misalign * 4: words misalign .. misalign + 3 if the
        // hardware honours the dword offset, words 0 .. 3 if it forces 16-byte alignment)
        uint32_t r[4];
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) uint32_t*>(&r[0])) : "v"(addr) : "memory");
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[65536] = r[0]; out[65537] = r[1]; out[65538] = r[2]; out[65539] = r[3]; }
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint32_t r0, r1, r2, r3;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) uint32_t*>(&r0)) : "v"(addr) : "memory");
        a0 ^= r0; a1 += r1; a2 ^= r2; a3 += r3;
        addr = (addr + 4096) & 0xFFFF;
        addr = (addr & ~0xFu) | ((addr + 0) & 0xFu);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, (256 * 256 + 4) * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int stride : {16, 20}) {            // 16: conflict-free rows of 16 B per lane; 20: the pitch does not matter for the question
        for (int mis = 0; mis < 4; ++mis) {
            hipLaunchKernelGGL(lds_read_kernel, dim3(256), dim3(256), 65536, 0, iters, mis, stride, out, cyc);
            hipError_t e = hipDeviceSynchronize();
            unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            uint32_t h[4]; hipMemcpy(h, out + 65536, 16, hipMemcpyDeviceToHost);
            int first = -1;
            for (int k = 0; k < 8; ++k) if (h[0] == (uint32_t)k * 2654435761u) first = k;
            printf("stride %2d B, byte offset %2d: %s, %.1f cycles per dependent ds_read_b128; lane 0 got words starting at %d (%s)\n", stride, mis * 4,
                   e == hipSuccess ? "ok" : hipGetErrorString(e), (double)c / iters, first, first == mis ? "dword offset honoured" : "address forced to 16-byte alignment");
            if (e != hipSuccess) return 1;
        }
    }
    return 0;
}
