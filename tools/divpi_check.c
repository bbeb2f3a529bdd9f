/* Exhaustive CPU check behind dist_ref.h's div_pi(): for every float32 a in [2^-13, pi]
 * (every value acosf can return other than 0) the multiply-and-correct sequence equals the
 * correctly rounded a / float32(pi).   gcc -O2 -fopenmp -ffp-contract=off -mfma tools/divpi_check.c -lm */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
int main(){
  const float pi=u2f(0x40490fdbu);
  const float inv=(float)(1.0/ (double)pi);
  long bad=0; long bad2=0;
  uint32_t hi=f2u(3.1415927f)+8;
  #pragma omp parallel for reduction(+:bad,bad2)
  for(uint64_t b=0x39000000u;b<=hi;b++){
    float a=u2f((uint32_t)b);
    float ref=a/pi;
    float q=a*inv; float r=fmaf(-q,pi,a); float q2=fmaf(r,inv,q);
    if(f2u(q2)!=f2u(ref)) bad++;
    float r2=fmaf(-q2,pi,a); float q3=fmaf(r2,inv,q2);
    if(f2u(q3)!=f2u(ref)) bad2++;
  }
  printf("inv=%a mismatches 1-step: %ld  2-step: %ld of %u\n",inv,bad,bad2,hi);
}
