#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 4000
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k(unsigned long long *out, int mode)
{
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    unsigned long long c0 = clock64();
    uint32_t acc = threadIdx.x;
    int s = mode;
    if (mode == 0) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("s_add_u32 %0, %0, 1\n") : "+s"(s) :: "scc"); }
    } else if (mode == 1) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("v_add_u32 %0, %0, 1\n") : "+v"(acc)); }
    } else if (mode == 2) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n") : "+s"(s) : "v"(acc) : "scc"); }
    } else if (mode == 3) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("s_branch 1f\n s_nop 0\n1:\n") ::); }
    } else if (mode == 4) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") :: "v"(0), "v"(acc) : "memory"); }
    } else if (mode == 5) {
        uint32_t a = 0;
        for (int i = 0; i < N; i++) { asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) :: "memory"); }
        acc += a;
    } else if (mode == 6) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n") : "+s"(s) :: "scc"); }
    } else if (mode == 7) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("s_cmp_lg_u32 %0, 12345\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n s_add_u32 %0, %0, 1\n") : "+s"(s) :: "scc"); }
    } else if (mode == 8) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("s_lshl_b64 s[20:21], 1, %0\n s_or_b64 s[22:23], s[22:23], s[20:21]\n s_bitcmp1_b64 s[22:23], %0\n s_addc_u32 %0, %0, 0\n") : "+s"(s) :: "s20","s21","s22","s23","scc"); }
    } else if (mode == 9) {
        for (int i = 0; i < N; i++) { asm volatile(REP16("v_mov_b32 %1, %0\n v_readfirstlane_b32 %0, %1\n") : "+s"(s), "+v"(acc)); }
    }
    unsigned long long t1 = wall_clock64();
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; out[2] = acc + s; }
}
int main()
{
    unsigned long long *d, h[3];
    hipMalloc(&d, 64);
    const char *names[] = {"16 dep s_add", "16 dep v_add", "16x(readlane+s_and)", "16 taken s_branch(+nop skipped)", "16x(ds_write+wait)", "16x(ds_read dep+wait)", "16x(cmp+nottaken br+add)", "16x(cmp+taken br+add)", "16x(4 salu b64)", "16x(v_mov+readfirstlane)"};
    for (int m = 0; m < 10; m++) {
        for (int r = 0; r < 2; r++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, m);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        }
        double ns = h[0] * 10.0;
        printf("%-34s %8.2f ns/iter  %8.2f ns per unit  (clock64 %llu per iter %.1f)\n", names[m], ns / N, ns / N / 16, h[1], (double)h[1] / N); fflush(stdout);
    }
    return 0;
}
