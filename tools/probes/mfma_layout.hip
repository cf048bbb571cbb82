#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// D[32][32] = A[32][K] * B[K][32]; each lane supplies a[], b[] as given arrays; dump acc
__global__ void k_f32(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}
__global__ void k_bf16(const u32x4* a, const u32x4* b, float* d) {
  int l = threadIdx.x;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[l]), __builtin_bit_cast(bf16x8, b[l]), c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}
static uint16_t bf(float f){ uint32_t u; memcpy(&u,&f,4); return u>>16; }
int main() {
  // ---- f32 32x32x2: assume A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
  {
    std::vector<float> A(32*2), B(2*32), a(64), b(64), d(64*16);
    for (int i=0;i<32;i++) for(int k=0;k<2;k++) A[i*2+k] = 1.0f + i + 100.0f*k;      // asymmetric
    for (int k=0;k<2;k++) for(int j=0;j<32;j++) B[k*32+j] = (k==0? 1.0f : 0.001f) * (1 + j);
    for (int l=0;l<64;l++){ a[l]=A[(l&31)*2+(l>>5)]; b[l]=B[(l>>5)*32+(l&31)]; }
    float *da,*db,*dd; hipMalloc(&da,256); hipMalloc(&db,256); hipMalloc(&dd,4096);
    hipMemcpy(da,a.data(),256,hipMemcpyHostToDevice); hipMemcpy(db,b.data(),256,hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_f32,1,64,0,0,da,db,dd); hipMemcpy(d.data(),dd,4096,hipMemcpyDeviceToHost);
    int bad=0;
    for (int l=0;l<64;l++) for(int r=0;r<16;r++){
      int row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31;
      float e = A[row*2+0]*B[0*32+col] + A[row*2+1]*B[1*32+col];
      if (fabsf(e-d[l*16+r])>1e-3f*fabsf(e)) { if(bad<5) printf("f32 mismatch lane %d reg %d got %f exp %f\n",l,r,d[l*16+r],e); bad++; }
    }
    printf("f32 32x32x2 layout check: %d mismatches\n", bad);
  }
  // ---- bf16 32x32x16: assume A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31]
  {
    std::vector<float> A(32*16), B(16*32), d(64*16);
    std::vector<uint16_t> a(64*8), b(64*8);
    for (int i=0;i<32;i++) for(int k=0;k<16;k++) A[i*16+k] = (float)((i+1) * ((k%4)+1)) * (k<8?1.0f:0.5f);
    for (int k=0;k<16;k++) for(int j=0;j<32;j++) B[k*32+j] = (float)((k==3||k==12)? (j+1) : 0);   // picks specific k's
    for (int l=0;l<64;l++) for(int e=0;e<8;e++){ int k=8*(l>>5)+e; a[l*8+e]=bf(A[(l&31)*16+k]); b[l*8+e]=bf(B[k*32+(l&31)]); }
    void *da,*db; float* dd; hipMalloc(&da,1024); hipMalloc(&db,1024); hipMalloc(&dd,4096);
    hipMemcpy(da,a.data(),1024,hipMemcpyHostToDevice); hipMemcpy(db,b.data(),1024,hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bf16,1,64,0,0,(const u32x4*)da,(const u32x4*)db,dd); hipMemcpy(d.data(),dd,4096,hipMemcpyDeviceToHost);
    int bad=0;
    for (int l=0;l<64;l++) for(int r=0;r<16;r++){
      int row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31;
      float e=0; for(int k=0;k<16;k++) e += A[row*16+k]*B[k*32+col];
      if (fabsf(e-d[l*16+r])>1e-2f*fabsf(e)+1e-3f) { if(bad<5) printf("bf16 mismatch lane %d reg %d got %f exp %f\n",l,r,d[l*16+r],e); bad++; }
    }
    printf("bf16 32x32x16 layout check: %d mismatches\n", bad);
  }
  return 0;
}
