// A caller of the whole-proof entry points with nothing but the C ABI (no Python, no torch): reads a circuit, its assignments and the
// scalars of a synthetic committer key from a file written by tests/test_varuna.py, pins the key, builds the index, proves, writes the
// proof and the verifier-key bytes.  What a Rust Varuna::prove_batch would do through FFI.
#include "aleo_mi355x.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>

static bool rd(FILE* f, void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
#define OK(call, what) do { int32_t rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s (%s)\n", what, aleo_mi355x_strerror(rc_), aleo_mi355x_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: varuna_prove_test <in> <out>\n"); return 2; }
  FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
  uint64_t h[8];          // n_constraints, n_public, n_private, max_degree, n_gamma, seed, instances, reserved
  if (!rd(f, h, sizeof h)) return 2;
  const uint64_t nc = h[0], npub = h[1], npriv = h[2], D = h[3], ng = h[4], seed64 = h[5], k = h[6], nv = npub + npriv;
  const aleo_mi355x::Seed seed = aleo_mi355x::Seed::from_u64(seed64), seed_next = aleo_mi355x::Seed::from_u64(seed64 + 1);
  if (nc > (1ull << 28) || nv > (1ull << 28) || D > (1ull << 28) || ng > 64 || k > 8) return 2;
  std::vector<uint8_t> gen(104), srs((D + 1 + ng) * 32);
  if (!rd(f, gen.data(), 104) || !rd(f, srs.data(), srs.size())) return 2;
  std::vector<uint32_t> rp[3], col[3]; std::vector<uint8_t> val[3]; aleo_mi355x_r1cs_matrix abc[3];
  for (int m = 0; m < 3; ++m) {
    uint64_t nnz; if (!rd(f, &nnz, 8) || nnz > (1ull << 28)) return 2;          // a malformed file must not turn into an allocation
    rp[m].resize(nc + 1); col[m].resize(nnz); val[m].resize(nnz * 32);
    if (!rd(f, rp[m].data(), (nc + 1) * 4) || !rd(f, col[m].data(), nnz * 4) || !rd(f, val[m].data(), nnz * 32)) return 2;
    abc[m].row_ptr = rp[m].data(); abc[m].col = col[m].data(); abc[m].val = val[m].data();
  }
  std::vector<std::vector<uint8_t>> z(k, std::vector<uint8_t>(nv * 32)); std::vector<const void*> zp(k);
  for (uint64_t i = 0; i < k; ++i) { if (!rd(f, z[i].data(), nv * 32)) return 2; zp[i] = z[i].data(); }
  fclose(f);

  OK(aleo_mi355x_init_device(0), "init");
  uint64_t key = 0, index = 0;
  OK(aleo_mi355x_bases_from_scalars(gen.data(), srs.data(), D + 1 + ng, &key), "bases_from_scalars");       // powers | hiding powers
  OK(aleo_mi355x_bases_precompute(key), "bases_precompute");
  OK(aleo_mi355x_varuna_index_build(&index, key, D, D + 1, 0, abc, nc, npub, npriv, (uint32_t)h[7]), "varuna_index_build");       // h[7]: domain flags
  uint8_t vk[12 * 48 + 64]; size_t vk_len = sizeof vk;
  OK(aleo_mi355x_varuna_index_vk(index, vk, &vk_len), "varuna_index_vk");
  std::vector<uint8_t> proof(2048); size_t len = proof.size();
  OK(aleo_mi355x_varuna_prove_indexed(index, zp.data(), k, seed.bytes, proof.data(), &len), "varuna_prove_indexed");
  size_t len2 = 16;                                          // a buffer that is too small is an error that reports the size needed
  if (aleo_mi355x_varuna_prove_indexed(index, zp.data(), k, seed.bytes, proof.data() + 1024, &len2) == 0 || len2 != len) { fprintf(stderr, "short buffer not refused\n"); return 1; }
  OK(aleo_mi355x_varuna_index_free(index), "varuna_index_free");
  if (aleo_mi355x_varuna_prove_indexed(index, zp.data(), k, seed.bytes, proof.data(), &len2) == 0) { fprintf(stderr, "freed index still usable\n"); return 1; }
  OK(aleo_mi355x_bases_unpin(key), "bases_unpin");
  FILE* o = fopen(argv[2], "wb"); if (!o) return 2;
  uint64_t l2[2] = {vk_len, len};
  fwrite(l2, 8, 2, o); fwrite(vk, 1, vk_len, o); fwrite(proof.data(), 1, len, o); fclose(o);
  // the same through the C++ mirror of ProvingKey::prove_batch (include/aleo_mi355x.hpp): same bytes, Display as proof1…
  {
    using namespace aleo_mi355x;
    auto ck = CommitterKey::from_scalars(*(const G1Affine*)gen.data(), (const BigInteger256*)srs.data(), D, ng);
    if (!ck.is_ok()) { fprintf(stderr, "CommitterKey: %s\n", ck.error.message().c_str()); return 1; }
    R1CS cs; cs.num_constraints = nc; cs.num_public = npub; cs.num_private = npriv;
    R1CSMatrix* mm[3] = {&cs.a, &cs.b, &cs.c};
    for (int m = 0; m < 3; ++m) { mm[m]->row_ptr = rp[m]; mm[m]->col = col[m]; mm[m]->val.resize(col[m].size()); memcpy(mm[m]->val.data(), val[m].data(), val[m].size()); }
    auto pk = ProvingKey::index(*ck.value, cs, (DomainPolicy)h[7]);
    if (!pk.is_ok()) { fprintf(stderr, "ProvingKey::index: %s\n", pk.error.message().c_str()); return 1; }
    std::vector<std::vector<BigInteger256>> za(k, std::vector<BigInteger256>(nv)); std::vector<const std::vector<BigInteger256>*> zs;
    for (uint64_t i = 0; i < k; ++i) { memcpy(za[i].data(), z[i].data(), nv * 32); zs.push_back(&za[i]); }
    auto pr = pk.value->prove_batch(zs, seed);
    if (!pr.is_ok() || pr.value->bytes.size() != len || memcmp(pr.value->bytes.data(), proof.data(), len)) { fprintf(stderr, "prove_batch differs from the raw call\n"); return 1; }
    auto str = pr.value->to_string();
    if (!str.is_ok() || str.value->rfind("proof1", 0) != 0) { fprintf(stderr, "Proof::to_string\n"); return 1; }
    auto vk2 = pk.value->verifying_key_bytes();
    if (!vk2.is_ok() || vk2.value->size() != vk_len || memcmp(vk2.value->data(), vk, vk_len)) { fprintf(stderr, "verifying_key_bytes\n"); return 1; }
    std::vector<BigInteger256> shorter(nv - 1); std::vector<const std::vector<BigInteger256>*> bad = {&shorter};
    if (pk.value->prove_batch(bad, seed).is_ok()) { fprintf(stderr, "wrong assignment length accepted\n"); return 1; }
    za[k - 1][nv - 1].l[0] ^= 1;                         // a witness that violates the circuit: refused with its own error code
    auto un = pk.value->prove_batch(zs, seed);
    if (un.is_ok() || !un.error.unsatisfied()) { fprintf(stderr, "unsatisfied assignment: expected ERR_UNSATISFIED, got %d\n", un.error.code); return 1; }
    za[k - 1][nv - 1].l[0] ^= 1;
    auto again = pk.value->prove_batch(zs, seed);
    if (!again.is_ok() || memcmp(again.value->bytes.data(), proof.data(), len)) { fprintf(stderr, "proof after a refused one differs\n"); return 1; }
    // a transaction with two functions: every instance under the first key, the first instance again under a second key of the same circuit —
    // Trace::prove_execution makes ONE proof; its bytes go to the output file for the restatement to compare
    auto pk2 = ProvingKey::index(*ck.value, cs, (DomainPolicy)h[7]);
    if (!pk2.is_ok()) { fprintf(stderr, "second ProvingKey::index\n"); return 1; }
    Trace trace;
    for (uint64_t i = 0; i < k; ++i) trace.insert_transition(*pk.value, za[i]);
    trace.insert_transition(*pk2.value, za[0]);
    if (trace.prove_fee(seed).is_ok()) { fprintf(stderr, "prove_fee accepted several transitions\n"); return 1; }
    auto ex = trace.prove_execution(seed);
    if (!ex.is_ok()) { fprintf(stderr, "Trace::prove_execution: %s\n", ex.error.message().c_str()); return 1; }
    Trace fee; fee.insert_transition(*pk.value, za[0]);
    auto fp = fee.prove_fee(seed_next), direct = pk.value->prove_batch({&za[0]}, seed_next);
    if (!fp.is_ok() || !direct.is_ok() || fp.value->bytes != direct.value->bytes) { fprintf(stderr, "Trace::prove_fee differs from prove_batch\n"); return 1; }
    // independent proofs in lockstep: the k-instance proof, the two-key execution, a fee, and a request with a violated witness in between — each
    // result what its own call gives, the bad one refused alone
    {
      std::vector<BigInteger256> broken = za[0]; broken[nv - 1].l[0] ^= 1;
      KeyedAssignments r0 = {{&*pk.value, zs}}, r1 = {{&*pk.value, zs}, {&*pk2.value, {&za[0]}}}, r2 = {{&*pk.value, {&broken}}}, r3 = {{&*pk.value, {&za[0]}}};
      auto many = prove_many({r0, r1, r2, r3}, {seed, seed, seed, seed_next});
      if (many.size() != 4 || !many[0].is_ok() || !many[1].is_ok() || !many[3].is_ok()) { fprintf(stderr, "prove_many: a good request failed\n"); return 1; }
      if (many[0].value->bytes.size() != len || memcmp(many[0].value->bytes.data(), proof.data(), len)) { fprintf(stderr, "prove_many[0] differs from prove_batch\n"); return 1; }
      if (many[1].value->bytes != ex.value->bytes) { fprintf(stderr, "prove_many[1] differs from Trace::prove_execution\n"); return 1; }
      if (many[2].is_ok() || !many[2].error.unsatisfied()) { fprintf(stderr, "prove_many[2]: expected ERR_UNSATISFIED\n"); return 1; }
      if (many[3].value->bytes != direct.value->bytes) { fprintf(stderr, "prove_many[3] differs from the fee proof\n"); return 1; }
    }
    // the queue in front of the lockstep call: six threads prove concurrently, each gets the proof its own call would have given, in fewer library calls than requests
    {
      ProvingQueue queue(8, std::chrono::microseconds(20000));
      std::vector<std::thread> th; std::vector<int> okv(6, 0);
      for (int t = 0; t < 6; ++t) th.emplace_back([&, t] {
        KeyedAssignments r = (t % 2) ? KeyedAssignments{{&*pk.value, {&za[0]}}} : KeyedAssignments{{&*pk.value, zs}};
        auto got = queue.submit(r, (t % 2) ? seed_next : seed).get();
        const auto& want = (t % 2) ? direct.value->bytes : pr.value->bytes;
        okv[t] = got.is_ok() && got.value->bytes == want;
      });
      for (auto& x : th) x.join();
      for (int t = 0; t < 6; ++t) if (!okv[t]) { fprintf(stderr, "ProvingQueue: request %d differs from its own call\n", t); return 1; }
      if (queue.calls() >= 6) { fprintf(stderr, "ProvingQueue: %zu library calls for 6 concurrent requests\n", queue.calls()); return 1; }
    }
    // the key sharded over "three devices" (the one card listed three times; min_points 0: every commitment goes through the shards): same proofs
    {
      Error e = ck.value->shard_over({0, 0, 0}, 0);
      if (e.code || ck.value->shards() != 3) { fprintf(stderr, "CommitterKey::shard_over: %s\n", e.message().c_str()); return 1; }
      auto sharded = pk.value->prove_batch(zs, seed);
      if (!sharded.is_ok() || sharded.value->bytes != pr.value->bytes) { fprintf(stderr, "proof against the sharded key differs\n"); return 1; }
      KeyedAssignments q0 = {{&*pk.value, zs}}, q1 = {{&*pk.value, {&za[0]}}};
      auto many = prove_many({q0, q1}, {seed, seed_next});
      if (many.size() != 2 || !many[0].is_ok() || !many[1].is_ok() || many[0].value->bytes != pr.value->bytes || many[1].value->bytes != direct.value->bytes) { fprintf(stderr, "lockstep proofs against the sharded key differ\n"); return 1; }
      auto pk3 = ProvingKey::index(*ck.value, cs, (DomainPolicy)h[7]);      // the index commitments through the shards too: same verifying key
      auto vk3 = pk3.is_ok() ? pk3.value->verifying_key_bytes() : Result<std::vector<uint8_t>>{std::nullopt, pk3.error};
      if (!vk3.is_ok() || vk3.value->size() != vk_len || memcmp(vk3.value->data(), vk, vk_len)) { fprintf(stderr, "verifying key of an index built against the sharded key differs\n"); return 1; }
      ck.value->unshard();
      auto back = pk.value->prove_batch(zs, seed);
      if (!back.is_ok() || back.value->bytes != pr.value->bytes || ck.value->shards() != 0) { fprintf(stderr, "proof after unshard differs\n"); return 1; }
    }
    FILE* o2 = fopen(argv[2], "ab"); if (!o2) return 2;
    uint64_t el = ex.value->bytes.size(); fwrite(&el, 8, 1, o2); fwrite(ex.value->bytes.data(), 1, el, o2); fclose(o2);
  }
  printf("ALL OK\n");
  return 0;
}
