#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <limits.h>
#define AVIF_SAMPLE_TRANSFORM_NEGATION 64
#define AVIF_SAMPLE_TRANSFORM_ABSOLUTE 65
#define AVIF_SAMPLE_TRANSFORM_NOT 66
#define AVIF_SAMPLE_TRANSFORM_SUM 128
#define AVIF_SAMPLE_TRANSFORM_DIFFERENCE 129
#define AVIF_SAMPLE_TRANSFORM_PRODUCT 130
#define AVIF_SAMPLE_TRANSFORM_QUOTIENT 131
#define AVIF_SAMPLE_TRANSFORM_AND 132
#define AVIF_SAMPLE_TRANSFORM_OR 133
#define AVIF_SAMPLE_TRANSFORM_XOR 134
#define AVIF_SAMPLE_TRANSFORM_POW 135
#define AVIF_SAMPLE_TRANSFORM_MIN 136
__host__ __device__ inline int32_t clamp32(int64_t v){ return v <= INT32_MIN ? INT32_MIN : (v >= INT32_MAX ? INT32_MAX : (int32_t)v); }
__host__ __device__ inline int bsr(int32_t a){
#ifdef __HIP_DEVICE_COMPILE__
  return a <= 0 ? 0 : 31 - __clz(a);
#else
  if (a<=0) return 0; int l=0; for (a>>=1; a; a>>=1) ++l; return l;
#endif
}
__host__ __device__ inline int32_t unaryOp(int32_t a, int type){ switch(type){case 64: return clamp32(-(int64_t)a); case 65: return a>=0?a:clamp32(-(int64_t)a); case 66: return ~a; default: return bsr(a);} }
__host__ __device__ inline int32_t binaryOp(int32_t l, int32_t r, int type){
  switch(type){ case 128: return clamp32((int64_t)l+r); case 129: return clamp32((int64_t)l-r); case 130: return clamp32((int64_t)l*r);
   case 131: return r==0?l:clamp32((int64_t)l/r); case 132: return l&r; case 133: return l|r; case 134: return l^r;
   case 135: { if(l==0||l==1) return l; if(l==-1) return (r%2==0)?1:-1; if(r==0) return 1; if(r==1) return l; if(r<0) return 0; int64_t res=l; for(int32_t i=1;i<r;++i){res*=l; if(res<INT32_MIN||res>INT32_MAX) return (l>0||r%2==0)?INT32_MAX:INT32_MIN;} return (int32_t)res; }
   case 136: return l<=r?l:r; default: return l<=r?r:l; } }
__global__ void k(const int32_t* L,const int32_t* R,int32_t* out,int n){ int i=blockIdx.x*256+threadIdx.x; if(i>=n) return; int op=i%14; out[i]= op<4? unaryOp(L[i],64+op) : binaryOp(L[i],R[i],128+op-4); }
int main(){ const int n=1<<20; int32_t *L=new int32_t[n],*R=new int32_t[n],*O=new int32_t[n]; unsigned s=1;
  int32_t sp[]={0,1,-1,2,-2,INT32_MIN,INT32_MAX,INT32_MIN+1,255,4095,65535,-39,128,3,31,32,33,-128};
  for(int i=0;i<n;++i){ s=s*1664525u+1013904223u; L[i]= (s&0x30000)? (int32_t)s>>( (s>>20)&31) : sp[(s>>8)%18]; s=s*1664525u+1013904223u; R[i]=(s&0x30000)? (int32_t)s>>((s>>20)&31) : sp[(s>>8)%18]; }
  int32_t *dL,*dR,*dO; hipMalloc(&dL,n*4);hipMalloc(&dR,n*4);hipMalloc(&dO,n*4); hipMemcpy(dL,L,n*4,hipMemcpyHostToDevice);hipMemcpy(dR,R,n*4,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k,dim3(n/256),dim3(256),0,0,dL,dR,dO,n); hipMemcpy(O,dO,n*4,hipMemcpyDeviceToHost);
  int bad=0; for(int i=0;i<n;++i){ int op=i%14; int32_t w= op<4? unaryOp(L[i],64+op):binaryOp(L[i],R[i],128+op-4); if(w!=O[i]){ if(bad<10) printf("op %d L=%d R=%d host=%d dev=%d\n",op<4?64+op:128+op-4,L[i],R[i],w,O[i]); ++bad;} }
  printf("mismatches %d of %d\n",bad,n); return 0; }
