// Throughput of three ways to write a large file in /dev/shm from page-resident buffers: one pwrite, pwrite from T threads,
// memcpy from T threads into a shared mapping.  g++ -O2 -fopenmp -o fs_bench fs_bench.cpp; ./fs_bench <path> <mode 0|1|2> <T>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <thread>
#include <atomic>
#include <cstdlib>
using namespace std;
static double now(){return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  const char*path=argv[1]; int mode=atoi(argv[2]); int T=atoi(argv[3]);
  size_t chunk=100u<<20; int nchunks=12;
  vector<char> buf(chunk); memset(buf.data(),'a',chunk);
  int fd=open(path,O_RDWR|O_CREAT|O_TRUNC,0644);
  double t0=now();
  uint64_t off=0;
  for(int c=0;c<nchunks;++c){
    if(mode==0){ // single pwrite
      pwrite(fd,buf.data(),chunk,off);
    } else if(mode==1){ // parallel pwrite
      #pragma omp parallel for num_threads(T) schedule(static,1)
      for(int t=0;t<T;++t){ size_t lo=chunk*t/T,hi=chunk*(t+1)/T; pwrite(fd,buf.data()+lo,hi-lo,off+lo);}
    } else if(mode==3){ // a second thread allocates the file's pages ahead (fallocate in 32 MB pieces), this one writes
      static std::thread *pre=nullptr; static std::atomic<uint64_t> ahead{0};
      if(!pre){ uint64_t total=(uint64_t)chunk*nchunks; pre=new std::thread([=]{ for(uint64_t o=0;o<total;o+=(32u<<20)){ fallocate(fd,0,o,32u<<20); ahead.store(o+(32u<<20)); } }); }
      pwrite(fd,buf.data(),chunk,off);
      if(c==nchunks-1){ pre->join(); }
    } else { // mmap
      ftruncate(fd,off+chunk);
      uint64_t a=off&~4095ull; size_t len=off+chunk-a;
      char*m=(char*)mmap(nullptr,len,PROT_READ|PROT_WRITE,MAP_SHARED,fd,a);
      if(m==MAP_FAILED){perror("mmap");return 1;}
      char*dst=m+(off-a);
      #pragma omp parallel for num_threads(T) schedule(static,1)
      for(int t=0;t<T;++t){ size_t lo=chunk*t/T,hi=chunk*(t+1)/T; memcpy(dst+lo,buf.data()+lo,hi-lo);}
      munmap(m,len);
    }
    off+=chunk;
  }
  double t1=now();
  close(fd);
  printf("mode %d T %d: %.2f GB/s\n",mode,T,off/(t1-t0)/1e9);
  unlink(path);
}
