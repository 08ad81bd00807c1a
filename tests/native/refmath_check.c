/* tests/native/refmath_check.c -- CPU build of tempestmodel_amd/csrc/tmx_refmath.h against the libm of this image,
 * bit for bit, on the argument ranges of the hot path and on wide ranges.  Built and run by
 * tests/test_abi_and_host_logic.py::test_refmath_is_glibc_bit_for_bit. */
#include <stdio.h>
#include <stdlib.h>
#include "tmx_refmath.h"   /* -I tempestmodel_amd/csrc */
static uint64_t s = 88172645463325252ull;
static double rnd(void){ s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0/9007199254740992.0); }
int main(int argc,char**argv){
	long n = argc>1? atol(argv[1]) : 20000000, bad_e=0,bad_l=0;
	for(long i=0;i<n;i++){
		double x;
		int m=i%4;
		if(m==0) x = -2.0 + 2.5*rnd();             /* exponent arguments of the Exner function */
		else if(m==1) x = -700.0 + 1400.0*rnd();
		else if(m==2) x = ldexp(rnd()-0.5, (int)(40*rnd())-30);
		else if(i%8==3) x = ldexp(rnd()-0.5, -(int)(1000*rnd())-40);   /* tiny, incl. |x| < 2^-54 */
		else x = 5.0*rnd()-2.5;
		if(tmx_ref_exp(x)!=exp(x)){ if(bad_e<5) printf("exp mismatch %a: %a vs %a\n",x,tmx_ref_exp(x),exp(x)); bad_e++; }
		/* the branch-free form the column kernel's assembly uses: |x| < 512 (tiny |x| included) */
		if(fabs(x)<512.0 && tmx_ref_exp_main(x, tmx_rm_exp_tab)!=exp(x)){ if(bad_e<5) printf("exp_main mismatch %a\n",x); bad_e++; }
		double y;
		if(m==0) y = 0.05 + 1.1*rnd();              /* R/p0*rho*theta */
		else if(m==1) y = 0.9375 + (1.0647-0.9375)*rnd();
		else if(m==2) y = ldexp(0.5+rnd(), (int)(2000*rnd())-1000);
		else y = 1e5*rnd();
		if(tmx_ref_log(y)!=log(y)){ if(bad_l<5) printf("log mismatch %a: %a vs %a\n",y,tmx_ref_log(y),log(y)); bad_l++; }
		/* branch-free form: positive, normal, finite arguments */
		if(tmx_ref_log_bf(y, tmx_rm_log_tab)!=log(y)){ if(bad_l<5) printf("log_bf mismatch %a\n",y); bad_l++; }
	}
	long bad_p=0;
	for(long i=0;i<n;i++){
		double x,y;
		int m=i%4;
		if(m==0){ x = 1e-3 + 1.2*rnd(); y = 0.2857142857142857 + 1e-3*(rnd()-0.5); }   /* (p/p0)^kappa of the Held-Suarez forcing */
		else if(m==1){ x = ldexp(0.5+rnd(), (int)(200*rnd())-100); y = 8.0*(rnd()-0.5); }
		else if(m==2){ x = 1e-8 + 50.0*rnd(); y = (rnd()<0.5)?0.875:((rnd()<0.5)?0.2046:0.525); }   /* Kessler's exponents */
		else { x = 0.9 + 0.2*rnd(); y = 2000.0*(rnd()-0.5); }
		if(tmx_ref_pow(x,y)!=pow(x,y)){ if(bad_p<5) printf("pow mismatch %a %a: %a vs %a\n",x,y,tmx_ref_pow(x,y),pow(x,y)); bad_p++; }
	}
	printf("n=%ld exp mismatches %ld log mismatches %ld pow mismatches %ld\n",n,bad_e,bad_l,bad_p);
	return (bad_e||bad_l||bad_p)?1:0;
}
