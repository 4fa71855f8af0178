#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template<int MODE> __global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters){
  float x[8]; f2 y[8];
  for(int i=0;i<8;++i){ x[i]=threadIdx.x*0.001f+i; y[i]=(f2){x[i],x[i]+1}; }
  f2 a2={a,a+0.1f}, b2={b,b*0.9f};
  for(int it=0;it<iters;++it){
    #pragma unroll
    for(int i=0;i<8;++i){
      if (MODE==0) x[i]=fmaf(x[i],a,b);
      else if (MODE==1) y[i]=__builtin_elementwise_fma(y[i],a2,b2);
      else if (MODE==2) x[i]=x[i]*a;           // v_mul
      else if (MODE==3) y[i]=y[i]*a2;          // v_pk_mul
      else if (MODE==4) x[i]= (x[i]>b)? a : x[i]; // cmp+cndmask
      else if (MODE==5) x[i]=__builtin_amdgcn_exp2f(x[i]);
      else if (MODE==6) x[i]=fminf(x[i],a)+b;  // min + add
    }
  }
  float s=0; for(int i=0;i<8;++i) s+=x[i]+y[i].x+y[i].y; out[blockIdx.x*256+threadIdx.x]=s;
}
int main(){ float* out; hipMalloc(&out, 4096*256*4); hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[]={"v_fma_f32","v_pk_fma_f32","v_mul_f32","v_pk_mul_f32","cmp+cndmask","v_exp_f32","min+add"};
  const int iters=4000; 
  for(int m=0;m<7;++m){ float best=1e9; for(int r=0;r<3;++r){ hipEventRecord(a);
    switch(m){case 0:k<0><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 1:k<1><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 2:k<2><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 3:k<3><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 4:k<4><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 5:k<5><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 6:k<6><<<4096,256>>>(out,1.0001f,0.5f,iters);break;}
    hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms;}
    double waveinstr = 4096.0*4*iters*8*((m==4||m==6)?2:1); // wave-level instructions
    double cyc = best*1e-3*2.4e9*1024/waveinstr;
    printf("%-14s %.3f ms  -> %.2f SIMD-cycles per wave64 instruction (at 2.4 GHz)\n", names[m], best, cyc);
  } return 0; }
