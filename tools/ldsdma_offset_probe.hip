// ldsdma_offset_probe.hip - does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// (band4_kernel_gen.py wants ONE m0 for the four pieces of a pre-tiled weight sub-tile.)   hipcc --offload-arch=gfx950 -o /tmp/p tools/ldsdma_offset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* __restrict__ src, unsigned* __restrict__ dst) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    unsigned voff = threadIdx.x * 16;
    unsigned base = (unsigned)(size_t)lds;      // LDS byte address of the array
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\ts_waitcnt vmcnt(0)" :: "v"(voff), "s"(src), "s"(base) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) dst[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    unsigned *s, *d;
    hipMalloc(&s, 8192 * 4); hipMalloc(&d, 4096 * 4);
    hipMemcpy(s, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(s, d);
    std::vector<unsigned> o(4096);
    hipMemcpy(o.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 4096; ++i) if (o[i] != 0xdeadbeefu) { first = i; break; }
    printf("first written LDS dword %d holds source dword %u (offset:2048 = 512 dwords): LDS destination %s by the immediate\n", first, first >= 0 ? o[first] : 0,
           first == 512 ? "MOVED" : (first == 0 ? "NOT moved" : "?"));
    return 0;
}
