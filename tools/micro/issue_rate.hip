// issue_rate.hip -- measured cost of the instruction kinds the ksw_extd2 kernels are made of (MI355X): wave64 throughput per SIMD of
// v_add_u32, v_pk_add_u16, v_pk_max_i16, v_pk_mul_lo_u16, v_mov_b32 DPP, s_add_u32, at 1..8 waves per SIMD.   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters)
{
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13, f = 17, g = 19, h = 23;
    unsigned sa = blockIdx.x, sb = 5, sc = 9, sd = 2;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 2) { REP16(asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 3) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %4\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (KIND == 4) { REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 5) { REP16(asm volatile("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4" : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) : "s"(7u) : "scc");) }
        if (KIND == 6) { REP16(asm volatile("v_add_u32 %0, %0, %3\n s_add_u32 %2, %2, %4\n v_add_u32 %1, %1, %3\n s_add_u32 %2, %2, %4" : "+v"(a), "+v"(b), "+s"(sa) : "v"(e), "s"(7u) : "scc");) }
        if (KIND == 7) { REP16(asm volatile("v_pk_lshlrev_b16 %0, 8, %0 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %1, 8, %1 op_sel_hi:[0,1]\n v_perm_b32 %2, %2, %3, %4\n v_alignbit_b32 %3, %3, %2, 16" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + sa + sb + sc + sd;
}
template <int KIND> void run(const char *name, int instr_per_iter)
{
    unsigned *out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {                    // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
        const int blocks = 256 * wps;
        k<KIND><<<blocks, 256>>>(out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0); k<KIND><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * instr_per_iter, waves_per_simd = wps;
        // cycles at ~2.4 GHz per (wave-instruction) per SIMD
        printf("%-28s %d waves/SIMD: %.3f ms  -> %.2f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wps, ms, ms * 1e6 / (instr_per_wave * waves_per_simd),
               ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4);
    }
}
int main()
{
    run<0>("v_add_u32", 64); run<1>("v_pk_add_u16", 64); run<2>("v_pk_max_i16", 64); run<3>("v_pk_mul_lo_u16", 64); run<4>("v_mov_b32_dpp wave_shr", 64);
    run<5>("s_add_u32", 64); run<6>("v_add_u32 + s_add_u32 mix", 64); run<7>("pk shifts / perm / alignbit", 64);
    return 0;
}
