"""Inputs on which libstdc++'s std::sort runs out of its depth limit and switches to heapsort (McIlroy's adversary:
"A Killer Adversary for Quicksort", 1999): keys are decided only while the sort compares them, always so that the pivot
ends up among the smallest of its range.  The device-side reproduction of the builder's sort must notice that case and
hand the sort to the host; tests feed these keys in as centroid coordinates.

    python tests/golden/make_sort_killer.py        # writes tests/golden/sort_killer.npz (needs g++)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

SOURCE = r"""
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv)
{
    const int n = atoi(argv[1]);
    const int gas = n - 1;
    std::vector<int> val(n, gas), item(n);
    int solid = 0, candidate = 0;
    for (int i = 0; i < n; ++i) item[i] = i;
    std::sort(item.begin(), item.end(), [&](int x, int y) {
        if (val[x] == gas && val[y] == gas) val[x == candidate ? x : y] = solid++;
        if (val[x] == gas) candidate = x;
        else if (val[y] == gas) candidate = y;
        return val[x] < val[y];
    });
    /* replay on the fixed keys, counting comparisons */
    std::vector<int> again(n);
    for (int i = 0; i < n; ++i) again[i] = i;
    long long compares = 0;
    std::sort(again.begin(), again.end(), [&](int x, int y) { ++compares; return val[x] < val[y]; });
    fprintf(stderr, "n %d compares %lld\n", n, compares);
    for (int i = 0; i < n; ++i) printf("%d\n", val[i]);
    return 0;
}
"""


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "killer.cpp"), os.path.join(tmp, "killer")
        open(src, "w").write(SOURCE)
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], check=True)
        for n in (600, 5000):
            r = subprocess.run([exe, str(n)], check=True, capture_output=True, text=True)
            sys.stderr.write(r.stderr)
            out[f"keys{n}"] = np.array(r.stdout.split(), np.int32)
            assert len(out[f"keys{n}"]) == n
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sort_killer.npz"), **out)


if __name__ == "__main__":
    main()
