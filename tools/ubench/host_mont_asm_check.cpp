#include "../../aleo_amd/csrc/host_field.hpp"
#include "host_mont_gen.h"      // python3 gen_host_mont_asm.py host_mont_gen.h; g++ -O3 -std=c++17 -mbmi2 -madx host_mont_asm_check.cpp
#include <cstdio>
#include <vector>
#include <chrono>
using namespace aleo_mi355x::host;
template <int N> int run(const char* nm) {
  using F = HFp<N>; using Pm = HParams<N>;
  uint64_t p7[N + 1]; for (int i = 0; i < N; ++i) p7[i] = Pm::P[i]; p7[N] = Pm::INV;
  uint64_t st = 12345 + N; auto next = [&]() { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
  int bad = 0; const int M = 200000; std::vector<F> A(M), B(M);
  for (int i = 0; i < M; ++i) { for (int k = 0; k < N; ++k) { A[i].l[k] = next(); B[i].l[k] = next(); } A[i] = F::reduce_lazy(A[i].l); B[i] = F::reduce_lazy(B[i].l); A[i].l[N-1] %= Pm::P[N-1]; B[i].l[N-1] %= Pm::P[N-1]; }
  // edge values
  std::memcpy(A[0].l, Pm::P, 8 * N); A[0].l[0] -= 1; B[0] = A[0]; A[1] = F::zero(); A[2] = F::one(); std::memcpy(B[3].l, Pm::P, 8 * N); B[3].l[0] -= 1;
  for (int i = 0; i < M; ++i) { F r; if (N == 6) mont_mul_asm6(r.l, A[i].l, B[i].l, p7); else mont_mul_asm4(r.l, A[i].l, B[i].l, p7); F w = fmul<N>(A[i], B[i]); if (!(r == w)) { if (bad < 3) printf("%s mismatch at %d\n", nm, i); ++bad; } }
  auto t0 = std::chrono::steady_clock::now(); F acc = A[5];
  for (int rep = 0; rep < 20; ++rep) for (int i = 0; i < M; ++i) { F r; if (N == 6) mont_mul_asm6(r.l, acc.l, B[i].l, p7); else mont_mul_asm4(r.l, acc.l, B[i].l, p7); acc = r; }
  auto t1 = std::chrono::steady_clock::now(); F acc2 = A[5];
  for (int rep = 0; rep < 20; ++rep) for (int i = 0; i < M; ++i) acc2 = fmul<N>(acc2, B[i]);
  auto t2 = std::chrono::steady_clock::now();
  printf("%s: %d bad of %d; asm %.1f ns, intrinsics %.1f ns per dependent product (%d)\n", nm, bad, M, std::chrono::duration<double, std::nano>(t1 - t0).count() / (20.0 * M), std::chrono::duration<double, std::nano>(t2 - t1).count() / (20.0 * M), (int)(acc == acc2));
  return bad;
}
int main() { return run<6>("Fq") + run<4>("Fr"); }
