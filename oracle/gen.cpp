// oracle/gen.cpp -- TEST INFRASTRUCTURE ONLY (see rnnt_oracle.c header).
//
// Deterministic input generators with the same streams the reference's test and
// timing harnesses use, so parity runs see the reference's own inputs:
//   acts   : std::mt19937 seeded 0, uniform_real_distribution<double>(0,1),
//            one draw per element, cast to float      (tests/random.cpp:4-21)
//   labels : std::mt19937 seeded 1, uniform_int_distribution<int>(1, A-1),
//            then two forced repeats when L >= 3      (tests/random.cpp:22-38)
// These are libstdc++ distributions, so they are reproduced by calling the
// same standard-library classes rather than by re-deriving them in numpy.
#include <cstddef>
#include <random>

extern "C" {

__attribute__((visibility("default")))
void oracle_gen_acts(float* out, size_t n) {
    std::mt19937 engine(0);
    std::uniform_real_distribution<> unit(0, 1);
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<float>(unit(engine));
}

__attribute__((visibility("default")))
void oracle_gen_acts_f64(double* out, size_t n) {
    std::mt19937 engine(0);
    std::uniform_real_distribution<> unit(0, 1);
    for (size_t i = 0; i < n; ++i) out[i] = unit(engine);
}

__attribute__((visibility("default")))
void oracle_gen_labels(int* out, int alphabet_size, int L) {
    std::mt19937 engine(1);
    std::uniform_int_distribution<> pick(1, alphabet_size - 1);
    for (int i = 0; i < L; ++i) out[i] = pick(engine);
    if (L >= 3) {  // guarantee repeated labels, as the reference does
        out[L / 2] = out[L / 2 + 1];
        out[L / 2 - 1] = out[L / 2];
    }
}

}  // extern "C"
