// Does v_mfma_f64_4x4x4_4b_f64 on gfx950 take DENORMAL A operands exactly and at full rate?
// Idea under test: an A value z in {0,1,2} written as the double with high word 0 and low word z (= z * 2^-1074, one v_bfe_u32
// instead of v_bfe_u32 + v_cvt_f64_u32), with B pre-scaled by 2^900 and the result scaled back by 2^174.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_check(const unsigned *z, const double *b, double *d_norm, double *d_den, int steps) {
  const int lane = threadIdx.x;
  double acc_n = 0.0, acc_d = 0.0;
  for (int s = 0; s < steps; s++) {
    const unsigned zz = z[s * 64 + lane];
    const double bb = b[s * 64 + lane];
    const double a_n = (double)zz;
    const double a_d = __hiloint2double(0, (int)zz);           // zz * 2^-1074
    acc_n = __builtin_amdgcn_mfma_f64_4x4x4f64(a_n, bb, acc_n, 0, 0, 0);
    acc_d = __builtin_amdgcn_mfma_f64_4x4x4f64(a_d, ldexp(bb, 900), acc_d, 0, 0, 0);
  }
  d_norm[lane] = acc_n;
  d_den[lane] = ldexp(acc_d, 174);
}

template <int DEN>
__global__ void __launch_bounds__(256, 1) k_speed(double *out, int iters, const unsigned *wsrc) {
  double acc[8][8], bf[8], af[8];
  for (int g = 0; g < 8; g++) {
    const unsigned zz = wsrc[(threadIdx.x + g * 64) & 1023] & 3u;
    af[g] = DEN ? __hiloint2double(0, (int)zz) : (double)zz;
    bf[g] = DEN ? ldexp(1.0 + 0.125 * g + threadIdx.x * 1e-6, 900) : 1.0 + 0.125 * g + threadIdx.x * 1e-6;
  }
  for (int g = 0; g < 8; g++) for (int h = 0; h < 8; h++) acc[g][h] = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 8; g++)
#pragma unroll
      for (int h = 0; h < 8; h++) acc[g][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[g], bf[h], acc[g][h], 0, 0, 0);
  }
  double s = 0;
  for (int g = 0; g < 8; g++) for (int h = 0; h < 8; h++) s += acc[g][h];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int steps = 4096;
  unsigned *hz = (unsigned *)malloc(4 * 64 * steps); double *hb = (double *)malloc(8 * 64 * steps);
  srand(1);
  for (int i = 0; i < 64 * steps; i++) { hz[i] = rand() % 3; hb[i] = ((double)rand() / RAND_MAX - 0.5) * exp(((double)rand() / RAND_MAX - 0.5) * 20.0); }
  unsigned *dz; double *db, *dn, *dd;
  CK(hipMalloc(&dz, 4 * 64 * steps)); CK(hipMalloc(&db, 8 * 64 * steps)); CK(hipMalloc(&dn, 512)); CK(hipMalloc(&dd, 512));
  CK(hipMemcpy(dz, hz, 4 * 64 * steps, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 8 * 64 * steps, hipMemcpyHostToDevice));
  k_check<<<1, 64>>>(dz, db, dn, dd, steps);
  double rn[64], rd[64];
  CK(hipMemcpy(rn, dn, 512, hipMemcpyDeviceToHost)); CK(hipMemcpy(rd, dd, 512, hipMemcpyDeviceToHost));
  int same = 0; double maxrel = 0;
  for (int i = 0; i < 64; i++) { same += rn[i] == rd[i]; if (rn[i] != 0) maxrel = fmax(maxrel, fabs(rn[i] - rd[i]) / fabs(rn[i])); }
  printf("denormal-A MFMA vs normal-A MFMA over %d accumulation steps: %d of 64 outputs bit-identical, max rel diff %.3e (sample %.17g vs %.17g)\n", steps, same, maxrel, rn[5], rd[5]);

  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, iters = 6000;
  double *out; unsigned *wsrc; CK(hipMalloc(&out, sizeof(double) * 256 * ncu)); CK(hipMalloc(&wsrc, 4096)); CK(hipMemset(wsrc, 0x59, 4096));
  for (int den = 0; den < 2; den++) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { if (den) k_speed<1><<<ncu, 256>>>(out, iters, wsrc); else k_speed<0><<<ncu, 256>>>(out, iters, wsrc); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("%s A operands: %.3f ms, %.2f TFLOP/s\n", den ? "denormal" : "normal  ", ms, (double)ncu * 4 * iters * 64 * 512.0 / ms * 1e-9);
  }
  return 0;
}
