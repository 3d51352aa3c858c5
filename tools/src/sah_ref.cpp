// Full-sweep top-down SAH over primitive boxes (the textbook builder): prints sum(A_inner)/A_root and
// sum(A_leaf * n)/A_root, a yardstick for tools/sah_cost.py.   usage: sah_ref boxes.bin [max_leaf]
// boxes.bin = n x {lo[3], hi[3]} float32.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
struct Box { float lo[3], hi[3]; };
static double area(const double lo[3], const double hi[3]) {
    double d[3] = {std::max(hi[0] - lo[0], 0.0), std::max(hi[1] - lo[1], 0.0), std::max(hi[2] - lo[2], 0.0)};
    return d[0] * d[1] + d[1] * d[2] + d[0] * d[2];
}
static std::vector<Box> B;
static double sum_inner = 0, sum_leaf = 0; static size_t n_inner = 0; static int max_leaf = 1;
static void grow(double lo[3], double hi[3], const Box &b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)b.lo[a]); hi[a] = std::max(hi[a], (double)b.hi[a]); } }
static void build(std::vector<uint32_t> &idx, size_t a, size_t b, bool root) {
    const size_t n = b - a;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = a; i < b; i++) grow(lo, hi, B[idx[i]]);
    const double A = area(lo, hi);
    if (n == 1) { sum_leaf += A; return; }
    double best = 1e300; int best_ax = -1; size_t best_k = 0;
    std::vector<double> right(n);
    for (int ax = 0; ax < 3; ax++) {
        std::sort(idx.begin() + a, idx.begin() + b, [&](uint32_t x, uint32_t y) { return B[x].lo[ax] + B[x].hi[ax] < B[y].lo[ax] + B[y].hi[ax]; });
        double rl[3] = {1e300, 1e300, 1e300}, rh[3] = {-1e300, -1e300, -1e300};
        for (size_t i = n; i-- > 1;) { grow(rl, rh, B[idx[a + i]]); right[i] = area(rl, rh); }
        double ll[3] = {1e300, 1e300, 1e300}, lh[3] = {-1e300, -1e300, -1e300};
        for (size_t k = 1; k < n; k++) {
            grow(ll, lh, B[idx[a + k - 1]]);
            const double c = area(ll, lh) * (double)k + right[k] * (double)(n - k);
            if (c < best) { best = c; best_ax = ax; best_k = k; }
        }
    }
    if ((int)n <= max_leaf && (double)n * A <= A + best) { sum_leaf += A * (double)n; return; }
    if (!root) sum_inner += A; else sum_inner += A;
    n_inner++;
    std::sort(idx.begin() + a, idx.begin() + b, [&](uint32_t x, uint32_t y) { return B[x].lo[best_ax] + B[x].hi[best_ax] < B[y].lo[best_ax] + B[y].hi[best_ax]; });
    build(idx, a, a + best_k, false);
    build(idx, a + best_k, b, false);
}
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (argc > 2) max_leaf = atoi(argv[2]);
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); size_t n = ftell(f) / sizeof(Box); fseek(f, 0, SEEK_SET);
    B.resize(n); if (fread(B.data(), sizeof(Box), n, f) != n) return 1; fclose(f);
    std::vector<uint32_t> idx(n); std::iota(idx.begin(), idx.end(), 0u);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (const Box &b : B) grow(lo, hi, b);
    const double root = area(lo, hi);
    build(idx, 0, n, true);
    printf("full-sweep SAH (max_leaf %d): inner %zu  sum(A_inner)/A_root %.2f  sum(A_leaf * n)/A_root %.2f\n", max_leaf, n_inner, sum_inner / root, sum_leaf / root);
    return 0;
}
