#include <chrono>
#include <cstdio>
#include <random>
#include <utility>
#include <vector>
#include "../guetzli_amd/host/lazy_sort.h"
typedef std::pair<int,float> E;
struct Less { bool operator()(const E&a,const E&b) const {return a.second<b.second;} };
int main(int argc,char**argv){
  size_t n=1800000; std::mt19937 rng(1);
  std::vector<E> v(n); for(size_t i=0;i<n;++i) v[i]=E(i,(float)(rng()%100000)/7);
  for (size_t thr : {(size_t)1<<30, (size_t)1<<17}) {
    double best=1e9;
    for(int rep=0;rep<5;++rep){ std::vector<E> w=v; auto t0=std::chrono::steady_clock::now();
      guetzli_amd::LazySorted<E,Less> s(w.data(),n,Less(),-1,thr); volatile float x=s[6500].second; (void)x;
      double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count(); if(dt<best)best=dt;}
    printf("threshold %zu: prefix 6500 of %zu: %.2f ms (pool %d)\n",thr,n,best*1e3,guetzli_amd::WorkerPool::Get().size());
  }
}
