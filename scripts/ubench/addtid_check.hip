// What ds_read_addtid_b32 reads on gfx950 (development aid): LDS address = instruction offset + M0 + 4 x lane.  Prints the
// addresses read for several M0 values, including bits above the documented 16: bit 16 counts, bit 31 does not.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out, unsigned m0v) {
    __shared__ unsigned tile[40960];   // 160 KB
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) tile[i] = i * 4;   // value = own byte address
    __syncthreads();
    unsigned r0, r1, r2;
    asm volatile("s_mov_b32 m0, %3\n s_nop 0\n ds_read_addtid_b32 %0\n ds_read_addtid_b32 %1 offset:1024\n ds_read_addtid_b32 %2 offset:65532\n s_waitcnt lgkmcnt(0)\n"
                 : "=v"(r0), "=v"(r1), "=v"(r2) : "s"(m0v) : "m0", "memory");
    out[threadIdx.x * 3 + 0] = r0; out[threadIdx.x * 3 + 1] = r1; out[threadIdx.x * 3 + 2] = r2;
    if (tile[threadIdx.x] == 0xffffffffu) out[0] = 1;
}
int main() {
    unsigned *d; hipMalloc(&d, 64 * 3 * 4); unsigned h[192];
    for (unsigned m0v : {0u, 256u, 51200u, 65280u, 0x80000000u + 512u, 0x10000u + 768u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, m0v);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("m0=%#x lane0: %u %u %u  lane5: %u %u %u\n", m0v, h[0], h[1], h[2], h[15], h[16], h[17]);
    }
    return 0;
}
