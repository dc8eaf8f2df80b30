// Host-only check of the spin pool (csrc/workers.hpp): every index is visited exactly once, for many jobs in a
// row, armed and disarmed, with and without helpers.  Built and run by tests/test_abi.py with plain g++.
#include "workers.hpp"
#include <cstdio>
#include <numeric>

static int check(SpinPool &pool, int64_t n, int64_t grain)
{
    std::vector<int> hits((size_t)n, 0);
    std::vector<int64_t> out((size_t)n, -1);
    pool.parallel_for(n, grain, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; i++) { hits[(size_t)i]++; out[(size_t)i] = i * 3; } });
    for (int64_t i = 0; i < n; i++) if (hits[(size_t)i] != 1 || out[(size_t)i] != i * 3) return 1;
    return 0;
}

int main()
{
    int bad = 0;
    for (int threads : {1, 2, 4}) {
        SpinPool pool(threads);
        bad += check(pool, 1000, 64);                 // not armed: inline
        for (int round = 0; round < 50; round++) {
            SpinPool::Armed guard(&pool);
            for (int job = 0; job < 40; job++) bad += check(pool, 1 + (job * 7919) % 5000, 1 + job % 97);
            bad += check(pool, 0, 8);
        }
        bad += check(pool, 12345, 100);               // disarmed again
    }
    printf(bad ? "FAIL %d\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
