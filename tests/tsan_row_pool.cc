// Host-only driver of mdbn_amd/csrc/row_pool.h for ThreadSanitizer (tests/test_tsan_row_pool.py): the spin / sleep hand-off
// between a gathering caller and the pool's workers, reused across many jobs of different sizes, with sleeping and with
// spinning workers, two pools at once (a feeder's dispatcher beside mdbn_host_gather_rows's private pool).
#include <cstdio>
#include <random>
#include <thread>
#include <vector>
#include "row_pool.h"

static int run(int threads, int jobs, bool let_sleep, unsigned seed)
{
    const int64_t n_rows = 257, cols = 96, ld = 100, ld_out = 96;
    std::vector<float> table(n_rows * ld), out(300 * ld_out);
    for (int64_t i = 0; i < n_rows; ++i)
        for (int64_t j = 0; j < cols; ++j) table[i * ld + j] = (float)(i * 1000 + j);
    mdbn_host::RowPool pool(threads - 1);
    std::mt19937 rng(seed);
    int bad = 0;
    for (int job = 0; job < jobs; ++job) {
        const int64_t n = 1 + rng() % 300;
        std::vector<int64_t> idx(n);
        for (auto& v : idx) v = rng() % n_rows;
        const bool poison = job % 17 == 16;                  // an out-of-range index must be reported, not read
        if (poison) idx[n / 2] = n_rows + 5;
        const bool ok = pool.gather(table.data(), n_rows, cols, ld, idx.data(), n, out.data(), ld_out);
        if (ok == poison) ++bad;
        for (int64_t r = 0; r < n; ++r) {
            if (poison && r == n / 2) continue;
            for (int64_t j = 0; j < cols; j += 31)
                if (out[r * ld_out + j] != table[idx[r] * ld + j]) ++bad;
        }
        if (let_sleep && job % 5 == 4) std::this_thread::sleep_for(std::chrono::milliseconds(4));   // past the spin window
    }
    return bad;
}

int main()
{
    int bad = 0;
    bad += run(1, 20, false, 1);
    bad += run(4, 120, false, 2);
    bad += run(4, 40, true, 3);
    std::thread other([&] { bad += run(3, 60, true, 4); });          // a second pool on another thread
    const int mine = run(5, 60, false, 5);
    other.join();
    bad += mine;
    std::printf("row pool: %d bad\n", bad);
    return bad ? 1 : 0;
}
