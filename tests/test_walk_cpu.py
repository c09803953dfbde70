"""CPU: the work distribution of the streaming decode kernels (csrc/fr_decode_shared.h: item_walk_round_robin /
item_walk_balanced) deals every tile to exactly one (workgroup, slot), whatever the tile count, grid and slot count -- the
header's own functions, compiled for the host by hipcc (no GPU, no HIP call) and enumerated.  A tile dealt twice or not at all
would be a wrong or missing vertex block on the GPU; the GPU parity tests see that only for the shapes they run."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, pkg

SRC = r'''
#include "%s/3dfacerecon_amd/csrc/fr_decode_shared.h"
#include <stdio.h>
#include <vector>
int main() {
    using namespace fr;
    long long cases = 0;
    const int grids[] = {1, 2, 7, 8, 16, 104, 256, 304};
    const int slotss[] = {2, 4, 6, 8, 12, 16};
    for (int mode = 0; mode < 2; mode++)
        for (int grid : grids)
            for (int slots : slotss)
                for (int tiles = 0; tiles <= 5000; tiles += (tiles < 70 ? 1 : 37 + tiles / 9)) {
                    std::vector<int> seen(tiles + 1, 0);
                    int most = 0, least = 1 << 30;
                    for (int b = 0; b < grid; b++) {
                        int per_cu = 0;
                        for (int slot = 0; slot < slots; slot++) {
                            const ItemWalk w = item_walk(mode, slot, slots, b, grid, tiles);
                            const int n = w.items();
                            int ct = n ? w.tile0() : tiles;
                            for (int it = 0; it < n; it++) {
                                if (ct < 0 || ct >= tiles) { printf("mode %%d grid %%d slots %%d tiles %%d: tile %%d out of range\n", mode, grid, slots, tiles, ct); return 1; }
                                seen[ct]++;
                                const int nt = w.next(it, ct, n, w.tile0());
                                if (nt < 0 || nt >= tiles) { printf("mode %%d grid %%d slots %%d tiles %%d: next %%d out of range\n", mode, grid, slots, tiles, nt); return 1; }
                                ct = nt;
                            }
                            per_cu += n;
                        }
                        most = per_cu > most ? per_cu : most;
                        least = per_cu < least ? per_cu : least;
                    }
                    for (int t = 0; t < tiles; t++)
                        if (seen[t] != 1) { printf("mode %%d grid %%d slots %%d tiles %%d: tile %%d dealt %%d times\n", mode, grid, slots, tiles, t, seen[t]); return 1; }
                    // the balanced walk's promise: the CUs' tile counts differ by at most two (one pair) -- by at most one when
                    // at least a whole round's worth of CUs ... is dealt; the round-robin walk may differ by a whole pair per slot pair
                    if (mode == 1 && tiles >= grid && most - least > 2) { printf("mode 1 grid %%d slots %%d tiles %%d: %%d vs %%d tiles per CU\n", grid, slots, tiles, most, least); return 1; }
                    cases++;
                }
    // the model's shape on an MI355X: 3,326 tiles, 256 CUs, 8 slots
    int cnt[2][256];
    for (int mode = 0; mode < 2; mode++)
        for (int b = 0; b < 256; b++) {
            cnt[mode][b] = 0;
            for (int slot = 0; slot < 8; slot++) cnt[mode][b] += item_walk(mode, slot, 8, b, 256, 3326).items();
        }
    int lo[2] = {99, 99}, hi[2] = {0, 0};
    for (int mode = 0; mode < 2; mode++)
        for (int b = 0; b < 256; b++) { lo[mode] = cnt[mode][b] < lo[mode] ? cnt[mode][b] : lo[mode]; hi[mode] = cnt[mode][b] > hi[mode] ? cnt[mode][b] : hi[mode]; }
    printf("ok %%lld cases; model shape tiles per CU: round robin %%d..%%d, balanced %%d..%%d\n", cases, lo[0], hi[0], lo[1], hi[1]);
    return 0;
}
'''


def test_item_walks_cover_every_tile_once(tmp_path):
    h = pkg("_lib")
    src = tmp_path / "walk.hip"
    src.write_text(SRC % ROOT)
    exe = tmp_path / "walk"
    r = subprocess.run([h._hipcc(), "--offload-arch=gfx950", "-O1", "-std=c++17", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "round robin 12..14, balanced 12..13" in r.stdout, r.stdout
