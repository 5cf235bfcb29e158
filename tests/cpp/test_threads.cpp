// test_threads.cpp -- concurrent callers, shaped after the reference's RNGExample (src/cpu/examples/RNGExample/
// StrongReproducibility/RNGExample.cpp:300-325 re-runs exsum and compares with !=; :549 calls it from pthreads
// with parallel=false).  Eight threads sum their own vectors repeatedly through the host-pointer API; every
// repetition of every thread must return the same bits.
#include "blas1.hpp"
#include "common.hpp"

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

int main()
{
    const int T = 8, N = 1 << 18, REP = 6;
    std::vector<std::vector<double>> v(T, std::vector<double>(N));
    srand(7);
    for (int t = 0; t < T; ++t) init_ill_cond(N, v[t].data(), 1e32);
    std::vector<double> first(T);
    for (int t = 0; t < T; ++t) first[t] = exsum(N, v[t].data(), 1, 0, 0);   // serial, superaccumulators only
    std::vector<int> bad(T, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (int r = 0; r < REP; ++r) {
                const double s = exsum(N, v[t].data(), 1, 0, (r & 1) ? 8 : 4, true, false);
                const double d = exdot(N, v[t].data(), 1, 0, v[(t + 1) % T].data(), 1, 0, 8, true);
                const double d2 = exdot(N, v[(t + 1) % T].data(), 1, 0, v[t].data(), 1, 0, 3);
                if (std::memcmp(&s, &first[t], 8) != 0 || std::memcmp(&d, &d2, 8) != 0) bad[t]++;
            }
        });
    for (auto &x : th) x.join();
    int nbad = 0;
    for (int t = 0; t < T; ++t) nbad += bad[t];
    printf(nbad == 0 ? "TestPassed; ALL OK!\n" : "TestFailed! (%d mismatches)\n", nbad);
    return nbad != 0;
}
