// Host-only check of the workspace layout's overlay argument (warp-transducer_amd/csrc/rnnt_host.h, make_layout), compiled by
// tests/test_workspace_layout.py with hipcc and run WITHOUT a GPU: for a grid of (maxT, maxU, N, lattice type)
//   1. the records of sample s end at or below the start of lattice block s      (nothing live is ever overwritten by an
//      in-order pass over the samples: coef_kernel's guard and coef_cell_kernel's groups both rest on it);
//   2. a group of `group` samples' records fits the head                          (coef_cell_kernel's group launches);
//   3. the row-scale array of the packed layout lies behind the record table and inside the blocks;
//   4. every array behind the blocks starts after them, 256-byte aligned, and `total` covers everything;
//   5. blocks are 256-byte aligned and the arrays inside a block do not overlap.
#include <cstdio>
#include <cstdlib>

#include "../../warp-transducer_amd/csrc/rnnt_host.h"

using namespace rnnt;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (failures < 20) { printf("FAIL %s: ", #cond); printf(__VA_ARGS__); printf("\n"); } ++failures; } } while (0)

int main() {
    const int Ts[] = {1, 2, 7, 30, 150, 200, 800, 1500, 4000};
    const int Us[] = {1, 2, 5, 21, 41, 64, 65, 130, 301, 1024};
    const int Ns[] = {1, 2, 3, 7, 8, 9, 16, 63, 64, 128, 1000, 1024, 70000};
    long long cases = 0;
    for (int lat : {4, 8})
        for (int T : Ts)
            for (int U : Us)
                for (int N : Ns) {
                    if (static_cast<long long>(T) * U > 0x7fffffffLL / 4) continue;
                    for (int joint = 0; joint < 2; ++joint) {
                        const Layout l = make_layout(T, U, N, lat, joint != 0);
                        const int Up = lat_stride(U);
                        const size_t block = lat_block(T, U, Up) * lat, rec1 = static_cast<size_t>(T) * U * 4 * lat, head = l.lp2;
                        const size_t cells = lat_rows(T, U) * static_cast<size_t>(Up);
                        ++cases;
                        CHECK(l.rowtab == 0 && head % 256 == 0 && block % 256 == 0, "T=%d U=%d N=%d lat=%d", T, U, N, lat);
                        CHECK(rec1 <= block, "a sample's records are larger than its block: T=%d U=%d lat=%d", T, U, lat);
                        CHECK(head >= rec1 || N == 0, "head < one sample's records: T=%d U=%d N=%d", T, U, N);
                        // 1. sample by sample (closed form: rec1 (s + 1) <= head + s block  <=  head >= rec1 and rec1 <= block; spot-check the ends)
                        for (size_t s : {size_t(0), size_t(N / 2), size_t(N - 1)})
                            CHECK(rec1 * (s + 1) <= head + s * block, "records of sample %zu reach into its own block: T=%d U=%d N=%d lat=%d", s, T, U, N, lat);
                        // 2. groups
                        CHECK(l.group >= 1 && l.group <= N, "group %d of N=%d", l.group, N);
                        CHECK(rec1 * static_cast<size_t>(l.group) <= head, "a group's records do not fit the head: T=%d U=%d N=%d group=%d", T, U, N, l.group);
                        if (l.group == N) CHECK(head >= rec1 * N, "one group but the table does not fit the head");
                        // 3. row scales
                        CHECK(l.rowscale >= rec1 * N && l.rowscale % 256 == 0, "rowscale inside the record table");
                        CHECK(l.rowscale + static_cast<size_t>(T) * U * N * lat <= l.offa, "rowscale reaches the arrays behind the blocks: T=%d U=%d N=%d", T, U, N);
                        // 4. what lies behind the blocks
                        const size_t blocks_end = head + block * N;
                        const size_t behind[] = {l.offa, l.offb, l.llf, l.llb, l.costs, l.coef_done, l.padflag, l.poison};
                        for (size_t o : behind) CHECK(o >= blocks_end && o % 256 == 0 && o < l.total, "array at %zu, blocks end at %zu, total %zu", o, blocks_end, l.total);
                        if (joint) CHECK(l.rowmax >= blocks_end && l.side > l.rowmax && l.wmat > l.side && l.wmat < l.total, "joint arrays");
                        // 5. inside a block
                        CHECK(l.logz - head == 2 * cells * lat && l.alpha - head == 3 * cells * lat && l.beta - head == 4 * cells * lat, "array offsets inside a block");
                        CHECK(5 * cells * lat + (Up + 64) * static_cast<size_t>(lat) <= block, "beta and its overshoot rows do not fit the block");
                        CHECK(lat_block(T, U, Up) % 2 == 0, "lp2's element stride is half the block: the block must be an even number of values");
                    }
                }
    // monotone in every argument (tests/test_abi.py checks it through the C entry; here across the group / head switch)
    for (int lat : {4, 8})
        for (int T : {150, 1500})
            for (int U : {21, 301}) {
                size_t prev = 0;
                for (int N = 1; N <= 600; ++N) {
                    const size_t t = make_layout(T, U, N, lat, false).total;
                    CHECK(t >= prev, "total shrinks from N=%d to N=%d (T=%d U=%d lat=%d)", N - 1, N, T, U, lat);
                    prev = t;
                }
            }
    printf("%lld layouts checked, %d failures\n", cases, failures);
    return failures == 0 ? 0 : 1;
}
