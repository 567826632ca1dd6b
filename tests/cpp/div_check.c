// Exhaustive check (all 2^23 significands, rcp seed off by -1, 0, +1 ulp and a few more) that
//   e = fma(-b, y0, 1); y1 = fma(e, y0, y0)  gives RN(1/b)
// and randomised check that  q0 = a*y1; r = fma(-b, q0, a); q = fma(r, y1, q0)  gives RN(a/b).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static inline float asf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static uint64_t s[2]={0x9E3779B97F4A7C15ull,0xD1B54A32D192ED03ull};
static inline uint64_t nxt(){uint64_t s1=s[0],s0=s[1];s[0]=s0;s1^=s1<<23;s[1]=s1^s0^(s1>>17)^(s0>>26);return s[1]+s0;}
int main(){
  long bad_rcp=0, bad_div=0, bad_div_allones=0; long ndiv=0;
  for(uint32_t m=0;m<(1u<<23);++m){
    float b=asf(0x3f800000u|m);           // [1,2)
    float yt=1.0f/b;                      // correctly rounded by IEEE division
    int ok_all=1;
    for(int d=-1;d<=1;++d){
      float y0=asf(asu(yt)+d);
      float e=fmaf(-b,y0,1.0f);
      float y1=fmaf(e,y0,y0);
      if(y1!=yt){ ++bad_rcp; ok_all=0; if(bad_rcp<10) printf("rcp miss m=%06x d=%d\n",m,d);}
    }
    float y=yt;
    for(int t=0;t<64;++t){
      uint64_t r=nxt();
      float a=asf(0x3f800000u|(uint32_t)(r&0x7fffff));
      if(t&1) a=asf(asu(a)+ (0x00800000u*((r>>40)%5)));  // other binades
      float q0=a*y; float rem=fmaf(-b,q0,a); float q=fmaf(rem,y,q0);
      ++ndiv;
      if(q!=a/b){ ++bad_div; if(bad_div<10) printf("div miss a=%a b=%a got %a want %a\n",a,b,q,a/b);}
    }
  }
  printf("rcp misses %ld of %d ; div misses %ld of %ld\n",bad_rcp,3*(1<<23),bad_div,ndiv);
  return 0;
}
