// Host-sanitizer driver of the C++ oracle leg: g++ -O1 -g -std=c++17 -fopenmp -fsanitize=address,undefined asan_driver.cpp -o /tmp/asan_driver && /tmp/asan_driver
// (six calls over growing and shrinking batch sizes, both arithmetic types, all four metrics; clean at the end of round 3).  Test infrastructure.
#include "dd3d_cpu.cpp"
#include <random>
int main() {
    std::mt19937 rng(7);
    const int J = 10, S = 49;
    auto rnd = [&](size_t n, double s) { std::vector<double> v(n); std::normal_distribution<double> d(0, s); for (auto& x : v) x = d(rng); return v; };
    auto W0 = rnd(27 * 3 * 128, 0.1), w1 = rnd(128, 1), b1 = rnd(128, 0.1), W1 = rnd((size_t)27 * 128 * 128, 0.03), w5 = rnd(128, 1), b5 = rnd(128, 0.1),
         Wc = rnd((size_t)J * 128 * S, 0.01), bc = rnd(J, 0.1);
    dd3d_cpu_net net{J, 1, 7, 7, W0.data(), w1.data(), b1.data(), W1.data(), w5.data(), b5.data(), Wc.data(), bc.data()};
    int64_t np = dd3d_cpu_num_params(&net);
    for (int round = 0; round < 6; ++round) {
        const int Br = (int[]){256, 3, 256, 1, 40, 256}[round], Bs = (int[]){10, 50, 1, 10, 2, 10}[round];
        std::vector<double> rc, rf, sc, sf;
        std::vector<int> ro{0}, so{0};
        std::uniform_int_distribution<int> npt(1, 250), px(0, 27);
        std::uniform_real_distribution<double> u(0, 27);
        for (int b = 0; b < Br; ++b) {
            int n = npt(rng);
            for (int i = 0; i < n; ++i) { double y = px(rng), x = px(rng); rc.insert(rc.end(), {0, y, x}); rf.insert(rf.end(), {0, y, x}); }
            ro.push_back(ro.back() + n);
        }
        for (int b = 0; b < Bs; ++b) {
            for (int i = 0; i < 250; ++i) { double y = u(rng), x = u(rng); sc.insert(sc.end(), {0, y, x}); sf.insert(sf.end(), {0, y, x}); }
            so.push_back(so.back() + 250);
        }
        double loss = 0;
        std::vector<double> grad(sf.size()), gwr(np), gws(np);
        int rcode = dd3d_cpu_class_step_f64(&net, rc.data(), rf.data(), ro.data(), Br, sc.data(), sf.data(), so.data(), Bs, round % J, round % 4, &loss, grad.data(), gwr.data(), gws.data());
        std::vector<float> rff(rf.begin(), rf.end()), sff(sf.begin(), sf.end()), g32(sf.size());
        std::vector<float> W0f(W0.begin(), W0.end()), w1f(w1.begin(), w1.end()), b1f(b1.begin(), b1.end()), W1f(W1.begin(), W1.end()), w5f(w5.begin(), w5.end()),
            b5f(b5.begin(), b5.end()), Wcf(Wc.begin(), Wc.end()), bcf(bc.begin(), bc.end());
        dd3d_cpu_net netf{J, 1, 7, 7, W0f.data(), w1f.data(), b1f.data(), W1f.data(), w5f.data(), b5f.data(), Wcf.data(), bcf.data()};
        float l32 = 0;
        int r32 = dd3d_cpu_class_step_f32(&netf, rc.data(), rff.data(), ro.data(), Br, sc.data(), sff.data(), so.data(), Bs, round % J, round % 4, &l32, g32.data(), nullptr, nullptr);
        printf("round %d: Br %d Bs %d rc %d loss %.9g | f32 rc %d loss %.7g\n", round, Br, Bs, rcode, loss, r32, (double)l32);
    }
    return 0;
}
