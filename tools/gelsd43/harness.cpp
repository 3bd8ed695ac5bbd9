// host build of csrc/gelsd43.h for tools/gelsd43/pin.py (staged comparison with the LAPACK routines of scipy's OpenBLAS)
#include "gelsd43.h"
using namespace mpc::gelsd43;
extern "C" {
void g_lartg(float f, float g, float *o) { lartg(f, g, o[0], o[1], o[2]); }
void g_las2(float f, float g, float h, float *o) { las2(f, g, h, o[0], o[1]); }
void g_lasv2(float f, float g, float h, float *o) { lasv2(f, g, h, o[0], o[1], o[2], o[3], o[4], o[5]); }
int g_bdsqr3(float *d, float *e, float *vt, float *c) { return bdsqr3(d, e, vt, c); }
int g_solve(const float *A, float *x) { return solve_ones(A, x); }
int g_solve_dbg(const float *A, float *x, float *dbg) { return solve_ones(A, x, dbg); }
}
