// Host sweep of nerf_sincosf against double precision; prints the max error in ulps.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include "../../nerf-pytorch_amd/csrc/sincos.h"
#include "../../nerf-pytorch_amd/csrc/sincos_table.h"

static const float kTable[2 * NERF_SINCOS_TABLE_ENTRIES] = {NERF_SINCOS_TABLE_VALUES};

static double ulp_of(float ref) {
    float a = fabsf(ref);
    if (a < 1e-30f) a = 1e-30f;
    int e; frexpf(a, &e);
    return ldexp(1.0, e - 24);
}
int main() {
    double worst_s = 0, worst_c = 0, worst_abs = 0; float ws = 0, wc = 0;
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) / 9007199254740992.0; };
    const double ranges[] = {1.0, 8.0, 100.0, 2100.0, 99999.0};
    for (double R : ranges)
        for (int i = 0; i < 4000000; ++i) {
            float x = (float)((rnd() * 2 - 1) * R);
            float s, c; nerf_sincosf(x, &s, &c);
            double rs = sin((double)x), rc = cos((double)x);
            double es = fabs(s - rs) / ulp_of((float)rs), ec = fabs(c - rc) / ulp_of((float)rc);
            if (es > worst_s) { worst_s = es; ws = x; }
            if (ec > worst_c) { worst_c = ec; wc = x; }
            double ea = fmax(fabs(s - rs), fabs(c - rc));
            if (ea > worst_abs) worst_abs = ea;
        }
    // exact multiples of powers of two times typical coordinates (positional-encoding arguments)
    for (int j = 0; j < 10; ++j)
        for (int i = -20000; i <= 20000; ++i) {
            float x = (float)(1 << j) * (float)(i * 1e-4);
            float s, c; nerf_sincosf(x, &s, &c);
            double ea = fmax(fabs(s - sin((double)x)), fabs(c - cos((double)x)));
            if (ea > worst_abs) worst_abs = ea;
        }
    printf("max_ulp_sin %.3f at %.9g\nmax_ulp_cos %.3f at %.9g\nmax_abs %.3e\n", worst_s, ws, worst_c, wc, worst_abs);
    // the table form (sine epilogues of the fused kernels): same sweep, |x| <= 1e5
    double worst_tab = 0; float wt = 0;
    for (double R : ranges)
        for (int i = 0; i < 4000000; ++i) {
            float x = (float)((rnd() * 2 - 1) * R);
            float s, c; nerf_sincosf_tab(x, kTable, &s, &c);
            double ea = fmax(fabs(s - sin((double)x)), fabs(c - cos((double)x)));
            if (ea > worst_tab) { worst_tab = ea; wt = x; }
        }
    for (int j = 0; j < 10; ++j)
        for (int i = -20000; i <= 20000; ++i) {
            float x = (float)(1 << j) * (float)(i * 1e-4);
            float s, c; nerf_sincosf_tab(x, kTable, &s, &c);
            double ea = fmax(fabs(s - sin((double)x)), fabs(c - cos((double)x)));
            if (ea > worst_tab) { worst_tab = ea; wt = x; }
        }
    {   // exact values on the axes and at zero
        float s, c;
        nerf_sincosf_tab(0.0f, kTable, &s, &c);
        if (s != 0.0f || c != 1.0f) { printf("table form: sincos(0) = %g %g\n", s, c); return 1; }
    }
    printf("table_max_abs %.3e at %.9g\n", worst_tab, wt);
    if (!(worst_tab < 1.0e-7)) return 1;
    float s, c; nerf_sincosf(3.0e7f, &s, &c);
    printf("large %.9g %.9g ref %.9g %.9g\n", s, c, sin(3.0e7), cos(3.0e7));
    return (worst_abs < 2.5e-7) ? 0 : 1;
}
