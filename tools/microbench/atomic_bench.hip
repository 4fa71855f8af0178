#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ unsigned rnd(unsigned x){ x^=x<<13; x^=x>>17; x^=x<<5; return x; }
template<int MODE> __global__ void k(unsigned* cnt, unsigned ncnt, unsigned* out, int per){
  unsigned g = blockIdx.x*256+threadIdx.x; unsigned s = g*2654435761u+12345u; unsigned acc=0;
  unsigned xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
  for(int i=0;i<per;++i){ s=rnd(s); unsigned t = s % ncnt;
    if (MODE==0) acc += atomicAdd(&cnt[t],1u);
    else if (MODE==1) atomicAdd(&cnt[t],1u);
    else if (MODE==2) acc += __hip_atomic_fetch_add(&cnt[t],1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE==3) acc += __hip_atomic_fetch_add(&cnt[xcd*ncnt+t],1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE==4) acc += atomicAdd(&cnt[xcd*ncnt+t],1u);
    else if (MODE==5) { acc += cnt[t]; }
    else if (MODE==6) { cnt[(size_t)g*per+i] = s; }
  }
  if (MODE!=1 && MODE!=6) out[g]=acc;
}
int main(){
  const unsigned ncnt=32400; const int N=2000000, per=4; unsigned *cnt,*out;
  hipMalloc(&cnt, (size_t)N*per*4 + 8*ncnt*4); hipMalloc(&out, N*4);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[]={"agent ret","agent noret","wg-scope ret","wg-scope ret xcd-private","agent ret xcd-private","plain load","plain store 4B"};
  for(int mode=0;mode<7;++mode){ float best=1e9;
    for(int rep=0;rep<5;++rep){ hipMemset(cnt,0,8*ncnt*4); hipEventRecord(a);
      dim3 grid((N+255)/256);
      switch(mode){case 0: k<0><<<grid,256>>>(cnt,ncnt,out,per);break; case 1:k<1><<<grid,256>>>(cnt,ncnt,out,per);break;
        case 2:k<2><<<grid,256>>>(cnt,ncnt,out,per);break; case 3:k<3><<<grid,256>>>(cnt,ncnt,out,per);break;
        case 4:k<4><<<grid,256>>>(cnt,ncnt,out,per);break; case 5:k<5><<<grid,256>>>(cnt,ncnt,out,per);break; case 6:k<6><<<grid,256>>>(cnt,ncnt,out,per);break;}
      hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
    printf("%-28s %.3f ms  %.1f Gops/s\n", names[mode], best, N*(double)per/best/1e6);
  }
  return 0;
}
