// Host accuracy harness for emei_amd/csrc/emei_math.h (the same header the HIP kernels include).
// Prints: max abs error of fast_sincos / fast_sincosf against long-double libm over several ranges,
// and the max relative error of the reciprocal-based division.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../emei_amd/csrc/emei_math.h"

int main() {
    std::mt19937_64 rng(12345);
    const double ranges[] = {0.8, 4.0, 100.0, 2000.0, 1.0e5, 1.0e6};
    for (double R : ranges) {
        std::uniform_real_distribution<double> U(-R, R);
        double worst = 0;
        for (int i = 0; i < 400000; ++i) {
            double x = U(rng), s, c;
            emei::fast_sincos(x, s, c);
            long double es = std::fabs((long double)s - sinl((long double)x));
            long double ec = std::fabs((long double)c - cosl((long double)x));
            if (es > worst) worst = (double)es;
            if (ec > worst) worst = (double)ec;
        }
        printf("f64 range %g maxabs %.3e\n", R, worst);
    }
    {   // table-assisted variant with a correctly rounded table (as emei_trig_table builds it)
        static emei::SinCosEntry tab[emei::kTrigTableSize];
        for (int k = 0; k < emei::kTrigTableSize; ++k) {
            long double a = 2.0L * 3.141592653589793238462643383279502884L * k / emei::kTrigTableSize;
            tab[k].s = (double)sinl(a), tab[k].c = (double)cosl(a);
        }
        for (double R : ranges) {
            std::uniform_real_distribution<double> U(-R, R);
            double worst = 0;
            for (int i = 0; i < 400000; ++i) {
                double x = U(rng), s, c;
                emei::fast_sincos_tab(x, tab, s, c);
                long double es = std::fabs((long double)s - sinl((long double)x));
                long double ec = std::fabs((long double)c - cosl((long double)x));
                if (es > worst) worst = (double)es;
                if (ec > worst) worst = (double)ec;
            }
            printf("tab range %g maxabs %.3e\n", R, worst);
        }
    }
    // near multiples of pi/2 (cancellation in the reduction)
    {
        double worst = 0;
        for (int k = -4000; k <= 4000; ++k)
            for (int j = -3; j <= 3; ++j) {
                double x = std::nextafter(k * 1.5707963267948966, j > 0 ? 1e9 : -1e9);
                x += j * 1e-9;
                double s, c;
                emei::fast_sincos(x, s, c);
                long double es = std::fabs((long double)s - sinl((long double)x));
                long double ec = std::fabs((long double)c - cosl((long double)x));
                if (es > worst) worst = (double)es;
                if (ec > worst) worst = (double)ec;
            }
        printf("f64 near-multiples maxabs %.3e\n", worst);
    }
    const float franges[] = {0.8f, 4.0f, 100.0f, 2000.0f, 3.0e4f};
    for (float R : franges) {
        std::uniform_real_distribution<float> U(-R, R);
        double worst = 0;
        for (int i = 0; i < 400000; ++i) {
            float x = U(rng), s, c;
            emei::fast_sincosf(x, s, c);
            double es = std::fabs((double)s - std::sin((double)x));
            double ec = std::fabs((double)c - std::cos((double)x));
            if (es > worst) worst = es;
            if (ec > worst) worst = ec;
        }
        printf("f32 range %g maxabs %.3e\n", (double)R, worst);
    }
    {
        std::uniform_real_distribution<double> D(0.25, 4.0), Nn(-100.0, 100.0);
        double worst = 0;
        for (int i = 0; i < 400000; ++i) {
            double d = D(rng), n = Nn(rng);
            double seed = (double)(1.0f / (float)d);  // a ~24-bit seed like a hardware rcp
            double r = emei::refine_rcp(d, seed);
            double q = emei::div_via_rcp(n, d, r);
            long double ref = (long double)n / (long double)d;
            double rel = (double)std::fabs(((long double)q - ref) / ref);
            if (rel > worst) worst = rel;
        }
        printf("div maxrel %.3e\n", worst);
    }
    // NaN / Inf propagate
    double s, c;
    emei::fast_sincos(INFINITY, s, c);
    printf("inf -> %s %s\n", std::isnan(s) ? "nan" : "num", std::isnan(c) ? "nan" : "num");
    emei::fast_sincos(NAN, s, c);
    printf("nan -> %s %s\n", std::isnan(s) ? "nan" : "num", std::isnan(c) ? "nan" : "num");
    return 0;
}
