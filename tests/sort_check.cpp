// rt::sortLikeStdSort against std::sort: the same permutation, element by element, on inputs where equal keys
// abound (so that the order std::sort leaves them in is visible), for several sizes and numbers of helper threads.
// Built by tests/Makefile with the system C++ compiler, run by tests/test_host_flatten.py.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../raytracing-in-one-weekend_amd/csrc/host_scene.h"

struct Item {
    float key;
    int tag;
    float ballast[6];  // the size of what the BVH builder sorts
};

int main() {
    auto less = [](const Item& a, const Item& b) { return a.key < b.key; };
    const size_t sizes[] = {1, 2, 16, 17, 1000, 32768, 40000, 300000, 1500000};
    const char* shapes[] = {"random", "few values", "one value", "ascending", "descending", "saw"};
    int failures = 0;
    for (size_t n : sizes) {
        for (int shape = 0; shape < 6; shape++) {
            std::mt19937 gen((unsigned)(n * 7 + (size_t)shape));
            std::vector<Item> input(n);
            for (size_t i = 0; i < n; i++) {
                float k = 0.0f;
                switch (shape) {
                    case 0: k = (float)gen() / 4294967296.0f; break;
                    case 1: k = (float)(gen() % 7); break;
                    case 2: k = 3.0f; break;
                    case 3: k = (float)(i / 3); break;
                    case 4: k = (float)((n - i) / 5); break;
                    default: k = (float)(i % 1000); break;
                }
                input[i].key = k;
                input[i].tag = (int)i;
            }
            std::vector<Item> expected = input;
            std::sort(expected.begin(), expected.end(), less);
            for (int threads : {1, 3, 8}) {
                std::vector<Item> got = input;
                std::atomic<int> spare(threads - 1);
                rt::sortLikeStdSort(got.data(), got.data() + n, less, spare);
                size_t bad = n;
                for (size_t i = 0; i < n; i++)
                    if (got[i].tag != expected[i].tag) {
                        bad = i;
                        break;
                    }
                if (bad != n || spare.load() != threads - 1) {
                    std::printf("MISMATCH n=%zu shape=%s threads=%d at %zu (spare %d)\n", n, shapes[shape], threads, bad, spare.load());
                    failures++;
                }
            }
        }
    }
    std::printf("%s\n", failures ? "FAILED" : "ok");
    return failures ? 1 : 0;
}
