// Host check of csrc/libstdcxx_sort_model.h against the real std::sort (libstdc++).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#include "libstdcxx_sort_model.h"

struct E { int cnt, x0, id; };
static bool lessE(const E& a, const E& b) { if (a.cnt < b.cnt) return true; if (a.cnt > b.cnt) return false; return a.x0 < b.x0; }

// median-of-3 killer (Musser) to force the heapsort fallback
static std::vector<int> killer(int n) {
    std::vector<int> v(n); int k = n / 2;
    for (int i = 1; i <= k; i++) { if (i % 2) { v[i - 1] = i; v[i] = k + i; } v[k + i - 1] = 2 * i; }
    return v;
}

int main() {
    std::mt19937 rng(12345);
    long cases = 0, bad = 0;
    for (int n = 0; n <= 1200; n += (n < 70 ? 1 : 37)) {
        for (int rep = 0; rep < 60; rep++) {
            std::vector<E> v(n);
            int mode = rep % 6;
            std::vector<int> kil = killer(n > 1 ? n : 2);
            for (int i = 0; i < n; i++) {
                int c, x;
                switch (mode) {
                    case 0: c = 2 + rng() % 3; x = (rng() % 4) * 90; break;          // heavy ties
                    case 1: c = 2 + rng() % 40; x = rng() % 720; break;
                    case 2: c = i; x = 0; break;                                      // sorted
                    case 3: c = n - i; x = 0; break;                                  // reversed
                    case 4: c = kil[i % kil.size()]; x = 0; break;                    // killer
                    default: c = 5; x = 7; break;                                     // all equal
                }
                v[i] = {c, x, i};
            }
            std::vector<E> a = v, b = v;
            std::sort(a.begin(), a.end(), lessE);
            orbx::libstdcxx_sort(b.data(), n, lessE);
            cases++;
            for (int i = 0; i < n; i++) if (a[i].id != b[i].id) { bad++; break; }
            // the data-parallel form of the partition step (what the quadtree kernel's waves execute)
            std::vector<E> c = v;
            orbx::libstdcxx_sort(c.data(), n, lessE, true);
            for (int i = 0; i < n; i++) if (a[i].id != c[i].id) { bad++; break; }
        }
    }
    printf("cases %ld bad %ld\n", cases, bad);
    return bad ? 1 : 0;
}
