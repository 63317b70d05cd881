// Micro-measurements that the cascade-kernel design leans on (run on the GPU box, prints one line per test):
//   * issue cost of the VALU instructions of a stump evaluation (v_add_u32, v_cvt_f32_i32, v_mul_f32, v_cndmask, v_add_f64)
//     at 1 / 2 / 4 / 8 wavefronts per SIMD, with the full EXEC mask and with only the lower 32 lanes enabled;
//   * LDS cycles of the gather forms: ds_read_b32 on consecutive / strided / random words, ds_read2_b32, ds_read_b64.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_cu tools/microbench_cu.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int ITER = 2048;

template <int OP, bool HALF>
__global__ __launch_bounds__(1024) void k_valu(unsigned long long* out, int* sink) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  const bool active = !HALF || (threadIdx.x & 63) < 32;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (active) {
    for (int i = 0; i < ITER; i++) {
      if (OP == 0) {
        asm volatile(
            "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
            "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(i));
      } else if (OP == 1) {
        asm volatile(
            "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
            "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
            : "v"(d0 * 0.0 + 1.0));
      } else if (OP == 2) {
        asm volatile(
            "v_cvt_f32_i32 %0, %8\n v_cvt_f32_i32 %1, %9\n v_cvt_f32_i32 %2, %10\n v_cvt_f32_i32 %3, %11\n"
            "v_cvt_f32_i32 %4, %12\n v_cvt_f32_i32 %5, %13\n v_cvt_f32_i32 %6, %14\n v_cvt_f32_i32 %7, %15\n"
            : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3), "=v"(f4), "=v"(f5), "=v"(f6), "=v"(f7)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
      } else if (OP == 3) {
        asm volatile(
            "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
            "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
            : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
            : "v"(1.0001f));
      } else if (OP == 4) {  // compare + select of a double (two v_cndmask) + f64 add: the vote of a stump
        asm volatile(
            "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %6, %7, vcc\n"
            "v_cmp_lt_f32 vcc, %3, %2\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %6, %7, vcc\n"
            "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %4, %5, vcc\n"
            : "+v"(a0), "+v"(a1)
            : "v"(f0), "v"(f1), "v"(a2), "v"(a3), "v"(a4), "v"(a5)
            : "vcc");
      } else if (OP == 5) {  // packed 16-bit integer add (the pair tile's corner arithmetic)
        asm volatile(
            "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
            "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(i));
      } else if (OP == 6) {  // conversion of one sign-extended half (SDWA operand)
        asm volatile(
            "v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
            "v_cvt_f32_i32_sdwa %2, sext(%10) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%11) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
            "v_cvt_f32_i32_sdwa %4, sext(%12) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %5, sext(%13) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
            "v_cvt_f32_i32_sdwa %6, sext(%14) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %7, sext(%15) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
            : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3), "=v"(f4), "=v"(f5), "=v"(f6), "=v"(f7)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
      } else if (OP == 7) {  // three-operand add
        asm volatile(
            "v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n"
            "v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(i));
      } else if (OP == 8) {  // v_cmpx + add under the narrowed EXEC + restore: the specialised vote
        asm volatile(
            "s_mov_b64 s[20:21], exec\n v_cmpx_gt_f32_e32 0x3f000000, %2\n v_add_u32_e32 %0, 0x1234567, %0\n s_mov_b64 exec, s[20:21]\n"
            "v_cmpx_gt_f32_e32 0x3f000000, %3\n v_add_u32_e32 %1, 0x1234567, %1\n s_mov_b64 exec, s[20:21]\n"
            "v_cmpx_gt_f32_e32 0x3f000000, %2\n v_add_u32_e32 %0, 0x1234567, %0\n s_mov_b64 exec, s[20:21]\n"
            "v_cmpx_gt_f32_e32 0x3f000000, %3\n v_add_u32_e32 %1, 0x1234567, %1\n s_mov_b64 exec, s[20:21]\n"
            : "+v"(a0), "+v"(a1)
            : "v"(f0), "v"(f1)
            : "vcc", "s20", "s21");
      }
    }
  }
  __syncthreads();  // every wavefront of the block is done (the oldest one wins arbitration and would finish early)
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123456789 || d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 1.5 ||
      f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 1.25f)
    sink[0] = 1;
}

// LDS gathers: every lane reads 8 words per iteration at (addr_k + iteration-independent offset); `mode` picks the form.
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(const int* __restrict__ addr, unsigned long long* out, int* sink) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  const int lane = threadIdx.x & 63;
  int a[8];
#pragma unroll
  for (int k = 0; k < 8; k++) a[k] = addr[k * 64 + lane] * 4;  // byte addresses
  __syncthreads();
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  int t0v = 0, t1v = 0, t2v = 0, t3v = 0, t4v = 0, t5v = 0, t6v = 0, t7v = 0;
  long long w0, w1, w2, w3, w4, w5, w6, w7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITER; i++) {
    if (MODE == 0) {  // 8 x ds_read_b32
      asm volatile(
          "ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n"
          "ds_read_b32 %4, %12\n ds_read_b32 %5, %13\n ds_read_b32 %6, %14\n ds_read_b32 %7, %15\n s_waitcnt lgkmcnt(0)\n"
          : "=&v"(t0v), "=&v"(t1v), "=&v"(t2v), "=&v"(t3v), "=&v"(t4v), "=&v"(t5v), "=&v"(t6v), "=&v"(t7v)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
      s0 += t0v; s1 += t1v; s2 += t2v; s3 += t3v; s4 += t4v; s5 += t5v; s6 += t6v; s7 += t7v;
    } else if (MODE == 1) {  // 4 x ds_read2_b32 (words a and a + 40)
      asm volatile(
          "ds_read2_b32 %0, %4 offset1:40\n ds_read2_b32 %1, %5 offset1:40\n ds_read2_b32 %2, %6 offset1:40\n ds_read2_b32 %3, %7 offset1:40\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
      s0 += (int)w0; s1 += (int)(w0 >> 32); s2 += (int)w1; s3 += (int)(w1 >> 32); s4 += (int)w2; s5 += (int)(w2 >> 32); s6 += (int)w3; s7 += (int)(w3 >> 32);
    } else if (MODE == 3) {  // 8 x ds_read_u16 at byte address = 2 * (pattern value): the pattern counts 16-bit entries
      asm volatile(
          "ds_read_u16 %0, %8\n ds_read_u16 %1, %9\n ds_read_u16 %2, %10\n ds_read_u16 %3, %11\n"
          "ds_read_u16 %4, %12\n ds_read_u16 %5, %13\n ds_read_u16 %6, %14\n ds_read_u16 %7, %15\n s_waitcnt lgkmcnt(0)\n"
          : "=&v"(t0v), "=&v"(t1v), "=&v"(t2v), "=&v"(t3v), "=&v"(t4v), "=&v"(t5v), "=&v"(t6v), "=&v"(t7v)
          : "v"(a[0] >> 1), "v"(a[1] >> 1), "v"(a[2] >> 1), "v"(a[3] >> 1), "v"(a[4] >> 1), "v"(a[5] >> 1), "v"(a[6] >> 1), "v"(a[7] >> 1));
      s0 += t0v; s1 += t1v; s2 += t2v; s3 += t3v; s4 += t4v; s5 += t5v; s6 += t6v; s7 += t7v;
    } else if (MODE == 4) {  // 4 x (ds_read_u16_d16 + ds_read_u16_d16_hi): two 16-bit entries into one register
      asm volatile(
          "ds_read_u16_d16 %0, %4\n ds_read_u16_d16_hi %0, %5\n ds_read_u16_d16 %1, %6\n ds_read_u16_d16_hi %1, %7\n"
          "ds_read_u16_d16 %2, %8\n ds_read_u16_d16_hi %2, %9\n ds_read_u16_d16 %3, %10\n ds_read_u16_d16_hi %3, %11\n s_waitcnt lgkmcnt(0)\n"
          : "+&v"(t0v), "+&v"(t1v), "+&v"(t2v), "+&v"(t3v)
          : "v"(a[0] >> 1), "v"(a[1] >> 1), "v"(a[2] >> 1), "v"(a[3] >> 1), "v"(a[4] >> 1), "v"(a[5] >> 1), "v"(a[6] >> 1), "v"(a[7] >> 1));
      s0 += t0v; s1 += t1v; s2 += t2v; s3 += t3v;
    } else {  // 8 x ds_read_b64 (addresses rounded down to 8 bytes by the host)
      asm volatile(
          "ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %10\n ds_read_b64 %3, %11\n"
          "ds_read_b64 %4, %12\n ds_read_b64 %5, %13\n ds_read_b64 %6, %14\n ds_read_b64 %7, %15\n s_waitcnt lgkmcnt(0)\n"
          : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "=&v"(w7)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
      s0 += (int)w0 + (int)(w0 >> 32); s1 += (int)w1; s2 += (int)w2; s3 += (int)w3; s4 += (int)w4; s5 += (int)w5; s6 += (int)w6; s7 += (int)w7;
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 == 123456789) sink[0] = 1;
}


// Do VALU work and LDS reads of the SAME wavefronts overlap? Per iteration 8 conflict-free ds_read_b32 (waited for at the end of the
// iteration) and NV independent VALU instructions of one kind: 0 = none, 1 = v_add_u32 (VOP2), 2 = v_pk_add_u16 (VOP3P), 3 = v_add3_u32 (VOP3).
// LDS = false drops the reads (VALU alone).
template <int KIND, int NV, bool LDS>
__global__ __launch_bounds__(1024) void k_mix(unsigned long long* out, int* sink) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  const int lane = threadIdx.x & 63;
  int a[8];
#pragma unroll
  for (int k = 0; k < 8; k++) a[k] = (k * 200 + lane) * 4;
  int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
  int t0v = 0, t1v = 0, t2v = 0, t3v = 0, t4v = 0, t5v = 0, t6v = 0, t7v = 0, acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITER; i++) {
    if (LDS)
      asm volatile(
          "ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n"
          "ds_read_b32 %4, %12\n ds_read_b32 %5, %13\n ds_read_b32 %6, %14\n ds_read_b32 %7, %15\n"
          : "=&v"(t0v), "=&v"(t1v), "=&v"(t2v), "=&v"(t3v), "=&v"(t4v), "=&v"(t5v), "=&v"(t6v), "=&v"(t7v)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
#pragma unroll
    for (int j = 0; j < NV / 4; j++) {
      if (KIND == 1)
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(i));
      if (KIND == 2)
        asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(i));
      if (KIND == 3)
        asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(i));
    }
    if (LDS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += t0v + t1v + t2v + t3v + t4v + t5v + t6v + t7v;
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc + v0 + v1 + v2 + v3 == 123456789) sink[0] = 1;
}

static double median_cycles(unsigned long long* d_out, int n) {
  std::vector<unsigned long long> h((size_t)n);
  CK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return (double)h[(size_t)n / 2];
}

int main() {
  unsigned long long* d_out;
  int *d_sink, *d_addr;
  const int blocks = 256;
  CK(hipMalloc(&d_out, sizeof(unsigned long long) * blocks));
  CK(hipMalloc(&d_sink, 4));
  CK(hipMalloc(&d_addr, 4 * 512));
  const char* opname[] = {"v_add_u32", "v_add_f64", "v_cvt_f32_i32", "v_mul_f32", "cmp+2cndmask(x8 of 9 instr)", "v_pk_add_u16", "v_cvt_f32_i32_sdwa sext", "v_add3_u32",
                          "v_cmpx+v_add (8 VALU of 8)"};
  for (int op = 0; op < 9; op++)
    for (int half = 0; half < 2; half++)
      for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        auto launch = [&](auto kern) {
          hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, d_sink);
          hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, d_sink);
          CK(hipDeviceSynchronize());
        };
        switch (op * 2 + half) {
          case 0: launch(k_valu<0, false>); break;
          case 1: launch(k_valu<0, true>); break;
          case 2: launch(k_valu<1, false>); break;
          case 3: launch(k_valu<1, true>); break;
          case 4: launch(k_valu<2, false>); break;
          case 5: launch(k_valu<2, true>); break;
          case 6: launch(k_valu<3, false>); break;
          case 7: launch(k_valu<3, true>); break;
          case 8: launch(k_valu<4, false>); break;
          case 9: launch(k_valu<4, true>); break;
          case 10: launch(k_valu<5, false>); break;
          case 11: launch(k_valu<5, true>); break;
          case 12: launch(k_valu<6, false>); break;
          case 13: launch(k_valu<6, true>); break;
          case 14: launch(k_valu<7, false>); break;
          case 15: launch(k_valu<7, true>); break;
          case 16: launch(k_valu<8, false>); break;
          case 17: launch(k_valu<8, true>); break;
        }
        const double cyc = median_cycles(d_out, blocks);
        const double instr_per_simd = (double)ITER * 8 * wps;  // one block per CU: wps wavefronts on each SIMD
        printf("VALU %-28s exec=%s waves/SIMD=%d : %.2f cycles per wave-instruction per SIMD\n", opname[op], half ? "lower32" : "full", wps,
               cyc / instr_per_simd);
      }
  // LDS address patterns (word index per lane, 8 slots)
  struct Pat {
    const char* name;
    std::vector<int> w;
  };
  std::vector<Pat> pats;
  auto mk = [&](const char* name, auto f) {
    Pat p{name, std::vector<int>(512)};
    for (int k = 0; k < 8; k++)
      for (int l = 0; l < 64; l++) p.w[(size_t)k * 64 + l] = f(k, l) & 8191;
    pats.push_back(p);
  };
  srand(7);
  mk("consecutive (lane)", [](int k, int l) { return k * 200 + l; });
  mk("stride 2 (2*lane)", [](int k, int l) { return k * 200 + 2 * l; });
  mk("random in 6080 words", [](int, int) { return rand() % 6080; });
  mk("all lanes one bank (32*lane)", [](int k, int l) { return k + 32 * l; });
  mk("pairs: 2*lane (b64 natural)", [](int k, int l) { return k * 200 + 2 * l; });
  mk("scattered, banks distinct per half", [](int k, int l) { return ((l * 7 + k * 3) % 32) + 32 * ((l * 13 + k) % 150); });
  const size_t n_common = pats.size();
  // ds_read_b64 at 4-byte granularity (mode 5: addresses as given, not rounded to 8 bytes): window pairs in one LDS plane
  mk("b64 pairs aligned (2*lane)", [](int k, int l) { return k * 200 + 2 * l; });
  mk("b64 pairs misaligned (2*lane+1)", [](int k, int l) { return k * 200 + 2 * l + 1; });
  mk("b64 mixed alignment (2*lane+(lane/7&1))", [](int k, int l) { return k * 200 + 2 * l + ((l / 7) & 1); });
  mk("b64 scattered, 8-byte banks distinct", [](int k, int l) { return 2 * ((l * 7 + k * 3) % 32) + 64 * ((l * 13 + k) % 75); });
  mk("b64 scattered, distinct, misaligned", [](int k, int l) { return 2 * ((l * 7 + k * 3) % 32) + 64 * ((l * 13 + k) % 75) + 1; });
  mk("b64 scattered, 4-byte class per half", [](int k, int l) { return ((l * 7 + k * 3) % 32) + 32 * ((l * 13 + k) % 150); });
  mk("b64 consecutive words (overlap)", [](int k, int l) { return k * 200 + l; });
  const size_t n_b64 = pats.size();
  // ds_read_b32 patterns of the pair tile's dense phase (mode 6): 16 pairs of one window row (even words) + the 16 below (odd row distance)
  mk("b32 pair-dense: 16 x stride 2 + 16 x stride 2 at +309", [](int k, int l) { return k * 7 + ((l >> 4) & 1) * 309 + ((l >> 5) * 16 + (l & 15)) * 2; });
  mk("b32 pair-dense, halves swapped (lanes 0-15,32-47 one row)", [](int k, int l) { return k * 7 + ((l >> 5) & 1) * 309 + (((l >> 4) & 1) * 16 + (l & 15)) * 2; });
  mk("b32 stride 2 in 16 lanes, +1 per 16 lanes", [](int k, int l) { return k * 7 + (l & 15) * 2 + ((l >> 4) & 1) + (l >> 5) * 64; });
  for (int mode = 0; mode < 7; mode++)
    for (size_t pi = 0; pi < pats.size(); pi++)
      for (int wps : {1, 4}) {
        const Pat& p = pats[pi];
        if (mode == 5 ? !(pi >= n_common && pi < n_b64) : mode == 6 ? pi < n_b64 : pi >= n_common) continue;
        std::vector<int> w = p.w;
        if (mode == 2)
          for (auto& x : w) x &= ~1;
        CK(hipMemcpy(d_addr, w.data(), 4 * 512, hipMemcpyHostToDevice));
        const int threads = 256 * wps;
        for (int rep = 0; rep < 2; rep++) {
          if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_addr, d_out, d_sink);
          if (mode == 1) hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_addr, d_out, d_sink);
          if (mode == 2) hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_addr, d_out, d_sink);
          if (mode == 3) hipLaunchKernelGGL(k_lds<3>, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_addr, d_out, d_sink);
          if (mode == 4) hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_addr, d_out, d_sink);
          if (mode == 6) hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_addr, d_out, d_sink);
          if (mode == 5) hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_addr, d_out, d_sink);
        }
        CK(hipDeviceSynchronize());
        const double cyc = median_cycles(d_out, blocks);
        const double n_instr = (double)ITER * (mode == 1 ? 4 : 8) * wps * 4;  // wave-instructions per CU
        const char* mname[] = {"ds_read_b32", "ds_read2_b32", "ds_read_b64", "ds_read_u16", "ds_read_u16_d16(+hi)", "ds_read_b64 (4B addr)", "ds_read_b32"};
        const double bytes_per_instr[] = {256.0, 512.0, 512.0, 128.0, 128.0, 512.0, 256.0};
        printf("LDS %-20s %-32s waves/CU=%2d : %.2f cycles per wave-instruction per CU (%.1f B/clk/CU)\n", mname[mode], p.name, wps * 4,
               cyc / n_instr, bytes_per_instr[mode] / (cyc / n_instr));
      }
  // VALU + LDS in the same wavefronts: 20 wavefronts per CU (the cascade kernel's residency), cycles per iteration per CU
  {
    const int threads = 1024;  // 16 wavefronts per CU (one block per CU)
    auto run = [&](const char* name, auto kern) {
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 8192 * 4 + 1024, 0, d_out, d_sink);
      CK(hipDeviceSynchronize());
      printf("MIX %-44s : %.1f cycles per iteration of the block's 16 wavefronts\n", name, median_cycles(d_out, blocks) / ITER);
    };
    run("8 ds_read_b32 alone", k_mix<0, 0, true>);
    run("8 v_add_u32 alone", k_mix<1, 8, false>);
    run("16 v_add_u32 alone", k_mix<1, 16, false>);
    run("8 v_pk_add_u16 alone", k_mix<2, 8, false>);
    run("8 v_add3_u32 alone", k_mix<3, 8, false>);
    run("8 ds_read_b32 + 8 v_add_u32", k_mix<1, 8, true>);
    run("8 ds_read_b32 + 16 v_add_u32", k_mix<1, 16, true>);
    run("8 ds_read_b32 + 8 v_pk_add_u16", k_mix<2, 8, true>);
    run("8 ds_read_b32 + 8 v_add3_u32", k_mix<3, 8, true>);
    run("8 ds_read_b32 + 4 v_pk_add_u16", k_mix<2, 4, true>);
  }
  return 0;
}
