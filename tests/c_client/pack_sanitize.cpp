// Host-only exercise of the weight packers under AddressSanitizer / UBSan.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "vad_amd.h"

static std::vector<float> rnd(size_t n, unsigned seed) {
    std::vector<float> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) % 2001 - 1000) / 1000.0f; }
    return v;
}

int main() {
    int fails = 0;
    // Net: every (lstm, large, hidden)
    for (int lstm = 0; lstm < 2; ++lstm)
        for (int large = 0; large < 2; ++large)
            for (int H : {1, 7, 30, 32, 33, 64, 90, 96}) {
                if (lstm && H > 32) continue;
                const int G = lstm ? 4 : 3;
                auto w_ih = rnd((size_t)G * H * 24, 1), w_hh = rnd((size_t)G * H * H, 2), b_ih = rnd(G * H, 3), b_hh = rnd(G * H, 4);
                auto l1w = rnd((size_t)26 * 30 * H, 5), l1b = rnd(26, 6), l2w = rnd(52, 7), l2b = rnd(2, 8);
                auto lw = rnd((size_t)2 * 30 * H, 9), lb = rnd(2, 10);
                const size_t n = vad_net_blob_floats(lstm, large, H);
                if (!n) { printf("no blob size for %d %d %d\n", lstm, large, H); ++fails; continue; }
                std::vector<float> blob(n);
                const int rc = vad_pack_net_weights(lstm, large, H, w_ih.data(), w_hh.data(), b_ih.data(), b_hh.data(), l1w.data(),
                                                    l1b.data(), l2w.data(), l2b.data(), lw.data(), lb.data(), blob.data(), n);
                if (rc) { printf("pack net %d %d %d -> %d (%s)\n", lstm, large, H, rc, vad_last_error()); ++fails; }
            }
    printf("net packers done, %d failures\n", fails);
    auto run_list = [&](const char* name, int large, std::vector<size_t> sizes, std::vector<int> positive, bool nick) {
        std::vector<std::vector<float>> store;
        std::vector<const float*> ptrs;
        for (size_t i = 0; i < sizes.size(); ++i) {
            store.push_back(rnd(sizes[i], 100 + (unsigned)i));
            if (positive[i]) for (auto& x : store.back()) x = (x < 0 ? -x : x) + 0.5f;
        }
        for (auto& v : store) ptrs.push_back(v.data());
        const size_t n = nick ? vad_nicknet_blob_floats(large) : vad_densenet_blob_floats(large);
        std::vector<float> blob(n);
        const int rc = nick ? vad_pack_nicknet_weights(large, ptrs.data(), (int)ptrs.size(), blob.data(), n)
                            : vad_pack_densenet_weights(large, ptrs.data(), (int)ptrs.size(), blob.data(), n);
        if (rc) { printf("%s large=%d -> %d (%s)\n", name, large, rc, vad_last_error()); ++fails; }
    };
    for (int large = 0; large < 2; ++large) {
        {   // DenseNet
            const int c0 = large ? 48 : 24, nl = large ? 8 : 6, g = large ? 4 : 3, c1 = c0 + nl * g;
            std::vector<size_t> sz; std::vector<int> pos;
            auto bn = [&](int c) { for (int k = 0; k < 4; ++k) { sz.push_back(c); pos.push_back(k == 3); } };
            sz.push_back((size_t)c0 * 24 * 6); pos.push_back(0); bn(c0);
            for (int i = 0; i < nl; ++i) { sz.push_back((size_t)g * (c0 + i * g) * 3); pos.push_back(0); bn(g); }
            sz.push_back((size_t)c0 * c1); pos.push_back(0); bn(c0);
            for (int i = 0; i < nl; ++i) { sz.push_back((size_t)g * (c0 + i * g) * 3); pos.push_back(0); bn(g); }
            sz.push_back((size_t)c1 * c1); pos.push_back(0); bn(c1);
            sz.push_back((size_t)2 * c1); pos.push_back(0);
            run_list("densenet", large, sz, pos, false);
        }
        {   // NickNet
            std::vector<int> ch = large ? std::vector<int>{24, 32, 28, 25, 18} : std::vector<int>{24, 20, 18, 16};
            const int H = large ? 29 : 21, layers = large ? 2 : 1;
            std::vector<size_t> sz; std::vector<int> pos;
            auto bn = [&](int c) { for (int k = 0; k < 4; ++k) { sz.push_back(c); pos.push_back(k == 3); } };
            for (size_t l = 0; l + 1 < ch.size(); ++l)
                for (int vg = 0; vg < 2; ++vg) { sz.push_back((size_t)ch[l + 1] * ch[l] * 3); pos.push_back(0); sz.push_back(ch[l + 1]); pos.push_back(0); bn(ch[l + 1]); }
            for (int gl = 0; gl < layers; ++gl) {
                const int in = gl == 0 ? ch.back() : H;
                sz.push_back((size_t)3 * H * in); sz.push_back((size_t)3 * H * H); sz.push_back(3 * H); sz.push_back(3 * H);
                for (int k = 0; k < 4; ++k) pos.push_back(0);
            }
            std::vector<int> w = large ? std::vector<int>{layers * H, 32, 18} : std::vector<int>{layers * H, 18};
            for (size_t st = 0; st + 1 < w.size(); ++st) { sz.push_back((size_t)w[st + 1] * w[st]); pos.push_back(0); sz.push_back(w[st + 1]); pos.push_back(0); bn(w[st + 1]); }
            sz.push_back((size_t)2 * w.back()); pos.push_back(0); sz.push_back(2); pos.push_back(0);
            run_list("nicknet", large, sz, pos, true);
        }
    }
    printf("all packers done, %d failures\n", fails);
    return fails;
}
