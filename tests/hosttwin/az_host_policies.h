// az_host_policies.h -- TEST INFRASTRUCTURE ONLY: the plain-C++ policies of the host twin (tests/hosttwin/azsp_host.cpp).
// Included from alpha_zero_amd/csrc/az_wave.h when AZ_HOST_TWIN_POLICIES is defined, i.e. never in the product build.
//   WaveHost : the wave programming model of az_wave.h as 64-iteration loops (same butterfly order in the sums)
#pragma once
struct WaveHost {
    static bool first() { return true; }
    template <class F> static void lanes(F&& f) {
        for (int l = 0; l < AZ_WAVE; ++l) f(l);
    }
    template <class F> static u64 ballot(F&& f) {
        u64 m = 0;
        for (int l = 0; l < AZ_WAVE; ++l)
            if (f(l)) m |= 1ull << l;
        return m;
    }
    static void sync() {}
    template <class F> static int argmax_first(F&& f) {
        double best = -1.0e300;
        int bi = 0x7fffffff;
        for (int l = 0; l < AZ_WAVE; ++l) {
            double s = -1.0e300;
            int idx = -1;
            f(l, s, idx);
            if (idx < 0) continue;
            if (s > best || (s == best && idx < bi)) {
                best = s;
                bi = idx;
            }
        }
        return bi;
    }
    static int bcast0(int v) { return v; }
    template <class T> static T uni(T v) { return v; }
    template <class F> static double sum_f64(F&& f) {
        // same butterfly order as the device so that non-integer sums (production noise) agree as well
        double v[AZ_WAVE];
        for (int l = 0; l < AZ_WAVE; ++l) v[l] = f(l);
        for (int o = 32; o > 0; o >>= 1) {
            double w[AZ_WAVE];
            for (int l = 0; l < AZ_WAVE; ++l) w[l] = v[l] + v[l ^ o];
            for (int l = 0; l < AZ_WAVE; ++l) v[l] = w[l];
        }
        return v[0];
    }
    template <class F> static int sum_i32(F&& f) {
        int v = 0;
        for (int l = 0; l < AZ_WAVE; ++l) v += f(l);
        return v;
    }
};

