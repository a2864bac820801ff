#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);}}while(0)
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(){
  size_t n=(size_t)1<<30;
  char *d0,*d1; CK(hipMalloc(&d0,n)); CK(hipMalloc(&d1,n));
  char *hp=(char*)aligned_alloc(4096,2*n); memset(hp,1,2*n);
  hipStream_t s0,s1; CK(hipStreamCreateWithFlags(&s0,hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1,hipStreamNonBlocking));
  for(int rep=0;rep<3;rep++){
    double t=now(); CK(hipMemcpyAsync(d0,hp,n,hipMemcpyHostToDevice,s0)); CK(hipStreamSynchronize(s0)); double up=now()-t;
    t=now(); CK(hipMemcpyAsync(hp+n,d1,n,hipMemcpyDeviceToHost,s1)); CK(hipStreamSynchronize(s1)); double dn=now()-t;
    t=now();
    std::thread th([&]{ CK(hipSetDevice(0)); CK(hipMemcpyAsync(hp+n,d1,n,hipMemcpyDeviceToHost,s1)); CK(hipStreamSynchronize(s1)); });
    CK(hipMemcpyAsync(d0,hp,n,hipMemcpyHostToDevice,s0)); CK(hipStreamSynchronize(s0));
    th.join(); double both=now()-t;
    // chunks of 32 MB from two threads
    t=now();
    std::thread th2([&]{ CK(hipSetDevice(0)); for(int i=0;i<32;i++) CK(hipMemcpyAsync(hp+n+i*(n/32),d1+i*(n/32),n/32,hipMemcpyDeviceToHost,s1)); CK(hipStreamSynchronize(s1)); });
    for(int i=0;i<32;i++) CK(hipMemcpyAsync(d0+i*(n/32),hp+i*(n/32),n/32,hipMemcpyHostToDevice,s0)); CK(hipStreamSynchronize(s0));
    th2.join(); double both32=now()-t;
    printf("pageable H2D %.1f GB/s D2H %.1f GB/s | two threads duplex %.1f ms (%.1f GB/s aggregate) | 32-piece %.1f ms\n",n/up/1e9,n/dn/1e9,both*1e3,2*n/both/1e9,both32*1e3);
  }
  return 0;
}
