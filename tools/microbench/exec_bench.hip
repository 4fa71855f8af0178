#include <hip/hip_runtime.h>
#include <stdio.h>
template<int MODE> __global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters){
  float x[8]; for(int i=0;i<8;++i) x[i]=threadIdx.x*0.001f+i;
  const int lane = threadIdx.x & 63;
  bool on = MODE==0 ? true : MODE==1 ? (lane<32) : MODE==2 ? (lane<16) : MODE==3 ? ((lane&1)==0) : (lane>=32);
  if (on) {
    for(int it=0;it<iters;++it){
      #pragma unroll
      for(int i=0;i<8;++i) x[i]=fmaf(x[i],a,b);
    }
  }
  float s=0; for(int i=0;i<8;++i) s+=x[i]; out[blockIdx.x*256+threadIdx.x]=s;
}
int main(){ float* out; hipMalloc(&out, 4096*256*4); hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[]={"all 64 lanes","lanes 0-31","lanes 0-15","even lanes","lanes 32-63"};
  const int iters=4000;
  for(int m=0;m<5;++m){ float best=1e9; for(int r=0;r<3;++r){ hipEventRecord(a);
    switch(m){case 0:k<0><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 1:k<1><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 2:k<2><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 3:k<3><<<4096,256>>>(out,1.0001f,0.5f,iters);break;case 4:k<4><<<4096,256>>>(out,1.0001f,0.5f,iters);break;}
    hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms;}
    printf("%-14s %.3f ms\n", names[m], best);
  } return 0; }
