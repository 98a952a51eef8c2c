/* g16_amd.h -- C ABI of the MI355X-native Groth16 (BN254) proving path for Circom circuits.
 *
 * Drop-in boundary for the proving path of arkworks-rs/circom-compat (ark-circom 0.5.0).  The
 * reference has no FFI of its own; the seam is the call
 *   Groth16::<Bn254, CircomReduction>::create_proof_with_reduction_and_matrices(
 *       &pk, r, s, &matrices, num_inputs, num_constraints, &full_assignment)
 * (reference benches/groth16.rs:52-60, src/zkey.rs:903-911) plus the loaders that feed it
 * (read_zkey, src/zkey.rs:53-60; R1CSFile::new, src/circom/r1cs_reader.rs:54-146).  Each entry
 * point below names the reference interface it replaces.  INTEGRATION.md shows the Rust
 * `extern "C"` binding a maintainer would add on the ark-circom side.
 *
 * Conventions
 *  - Field elements are 4 x u64 little-endian limbs in MONTGOMERY form (R = 2^256): exactly the
 *    in-memory form of ark_bn254::{Fr,Fq} (`x.0.0`) and the on-disk form of zkey points
 *    (src/zkey.rs:327-332).  "Fr" arguments (witness, r, s, CSR coefficients, h) are Montgomery.
 *  - G1 affine = x|y (64 bytes), G2 affine = x.c0|x.c1|y.c0|y.c1 (128 bytes), all-zero = point
 *    at infinity: the packed form deserialize_g1/g2 decode (src/zkey.rs:340-360).
 *  - Every function returns a g16_status; no exceptions cross the boundary; the caller owns all
 *    buffers; a ctx is not re-entrant (one call in flight per ctx, several ctxs allowed).
 *  - There is NO CPU fallback: without a usable HIP device g16_ctx_create fails with
 *    G16_ERR_NO_DEVICE.
 */
#ifndef G16_AMD_H
#define G16_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int g16_status;
enum {
  G16_OK = 0,
  G16_ERR_INVALID = 1,          /* bad argument / size mismatch                                  */
  G16_ERR_DOMAIN_TOO_LARGE = 2, /* SynthesisError::PolynomialDegreeTooLarge (qap.rs:31,66)       */
  G16_ERR_HIP = 3,              /* HIP runtime error, see g16_last_error                         */
  G16_ERR_NO_DEVICE = 4,        /* no gfx950 device visible: the product path refuses to run     */
  G16_ERR_IO = 5,               /* SerializationError / io error in a loader                     */
  G16_ERR_INTERNAL = 6
};

typedef struct g16_ctx g16_ctx;

/* Row-major sparse matrix = ConstraintMatrices::{a,b} (Vec<Vec<(Fr, usize)>>, src/zkey.rs:165-194)
 * flattened to CSR.  coeff: nnz x 4 u64, Montgomery.                                              */
typedef struct {
  const uint32_t* row_ptr; /* [num_constraints + 1] */
  const uint32_t* col;     /* [nnz] wire index      */
  const uint64_t* coeff;   /* [nnz][4]              */
  uint64_t nnz;
} g16_csr;

/* ProvingKey<Bn254> as read_zkey builds it (src/zkey.rs:103-133), packed arrays on the host.     */
typedef struct {
  uint32_t n_vars;      /* N: wires incl. the constant 1 (HeaderGroth.n_vars)                    */
  uint32_t n_public;    /* p (HeaderGroth.n_public); num_inputs = p + 1                          */
  uint32_t domain_size; /* n = len(h_query)                                                      */
  const uint8_t* a_query;    /* N x 64           (zkey section 5) */
  const uint8_t* b_g1_query; /* N x 64           (section 6)      */
  const uint8_t* b_g2_query; /* N x 128          (section 7)      */
  const uint8_t* l_query;    /* (N-p-1) x 64     (section 8)      */
  const uint8_t* h_query;    /* domain_size x 64 (section 9)      */
  uint8_t alpha_g1[64], beta_g1[64], delta_g1[64];
  uint8_t beta_g2[128], delta_g2[128];
} g16_key_desc;

typedef struct {
  int device;      /* HIP device ordinal                                                        */
  int rank, world; /* point-range shard of the MSMs owned by this ctx (world = 1: everything)   */
  int window_bits; /* MSM window c; <= 0: automatic                                              */
  int planes;      /* stored multiples 2^(c*D*j)P per point; <= 0: as many as fit (full = W)     */
  int dist_wm;     /* > 0: distribute the witness map too; the ctx then proves ONLY through the
                      g16_prove_dist_phase* calls (g16_prove answers G16_ERR_INVALID).  world = 1 with
                      dist_wm = 1 is the degenerate one-rank case of that API (every exchange copies onto
                      itself: what a one-process run of the host framework's collectives drives)      */
  int reduction;   /* G16_REDUCTION_CIRCOM (0, default) or G16_REDUCTION_LIBSNARK                  */
  int shard;       /* world > 1: G16_SHARD_AUTO (0), G16_SHARD_POINTS, G16_SHARD_BUCKETS (below)    */
  int fixed_tables; /* small keys: every MSM of g16_prove as table lookups + a tree sum (per point and
                      8-bit window the multiples 1..128: 256 KiB per G1 point, 512 KiB per G2 point) and
                      the finalisation's variable-base products as two more table MSMs -- ~20 launches
                      with ~21 dependent EC additions each instead of ~75 with bucket reductions.
                      0: automatic (single-device proving ctx, <= 2^14 points per query, tables within a
                      third of the free device memory); > 0: require it (creation fails where it cannot
                      apply); < 0: never.  Results are the same group elements either way.            */
} g16_options;

/* How the MSMs of one proof are cut over `world` ranks:
 *   POINTS   every query array by contiguous point range; a rank sorts and accumulates its n/world
 *            points with the window that suits n/world (more windows, a bucket set per rank);
 *   BUCKETS  the four witness-scalar queries (A, B1, B2, L -- the witness is resident on every rank
 *            anyway): every rank keeps ALL their points (sized for 288 GB: 18 GiB at 2^22, 72 GiB at
 *            2^24) and the single-GPU window; the sorted (bucket, point) list is cut into `world`
 *            equal runs of whole sort partitions, chosen on the device from the digit histogram, and
 *            rank g accumulates and reduces run g only.  Same additions per point as on one GPU,
 *            1/world of the bucket reduction per rank, and no bucket sums on the links: the only MSM
 *            traffic is the 1 KiB record per rank.  The H query stays cut by point range -- its
 *            scalars are born sharded (the distributed witness map leaves rank g its n / world
 *            evaluations), moving them would put the whole vector on every link.
 *   AUTO     POINTS.  One rank of 8 measured alone on an MI355X takes the same time either way
 *            (7.6 / 7.8 ms at 2^22, 21.0 / 21.4 ms at 2^24, profiles/r03_proj_*.json): the window a
 *            bucket-sharded rank saves is paid back by walking all n scalars to keep an eighth of the
 *            digits; POINTS holds 1/world of the key per device.  BUCKETS stays selectable (it is the
 *            better cut where the per-device key does not matter and the windows differ more).      */
enum { G16_SHARD_AUTO = 0, G16_SHARD_POINTS = 1, G16_SHARD_BUCKETS = 2 };

/* The R1CS -> QAP reduction (the `QAP` type parameter of ark_groth16::Groth16<E, QAP>):
 *   CIRCOM   = ark_circom::CircomReduction (reference src/circom/qap.rs:12-106): snarkjs keys (.zkey)
 *   LIBSNARK = ark_groth16::LibsnarkReduction, the default of `Groth16<Bn254>` used with
 *              arkworks-generated keys (reference tests/groth16.rs:9,25-35; README.md:69-74 explains
 *              why the two must not be mixed).  Its H query has domain_size - 1 points: pass it
 *              padded with the point at infinity (all-zero) to domain_size entries.
 * The matrices handed to g16_ctx_create hold A and B only (as read_zkey produces them,
 * src/zkey.rs:188-192); c_i = a_i * b_i is used where LibsnarkReduction evaluates C.w, which is
 * the same value for a satisfying assignment (g16_check_satisfied tests that).                     */
enum { G16_REDUCTION_CIRCOM = 0, G16_REDUCTION_LIBSNARK = 1 };

#define G16_PROOF_BYTES 256   /* A(64) | B(128) | C(64), affine */
#define G16_PARTIAL_BYTES 1024 /* A | B1 | B2 | L | H | s*A | r*B1: one rank's sums, XYZZ (x, y, zz, zzz;
                                  x/zz, y/zzz affine), Montgomery; G1 128 B, G2 256 B; zz = 0: infinity */

enum { G16_QUERY_A = 0, G16_QUERY_B1 = 1, G16_QUERY_L = 2, G16_QUERY_H = 3 };

/* Uploads the key and the matrices once, precomputes NTT tables and MSM point planes.
 * Replaces: holding `(ProvingKey<Bn254>, ConstraintMatrices<Fr>)` from read_zkey (src/zkey.rs:53-60)
 * across calls.  num_constraints = matrices.num_constraints (src/zkey.rs:171).
 * Sizes: every domain the reference accepts is accepted -- G16_ERR_DOMAIN_TOO_LARGE exactly where
 * CircomReduction raises PolynomialDegreeTooLarge (num_constraints + num_inputs > 2^27: no 2n-domain,
 * src/circom/qap.rs:30-32,63-68).  The point planes are planned against the device memory that is free
 * at the call: full precomputation up to 2^25 constraints on a 288 GB device, fewer planes (more bucket
 * sets per MSM, same results) above; G16_ERR_INTERNAL "does not fit this device's memory even with
 * one plane per point" only when nothing fits.  opt->planes > 0 overrides the plan.               */
g16_status g16_ctx_create(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                          uint32_t num_constraints, const g16_options* opt, g16_ctx** out);
void g16_ctx_destroy(g16_ctx* ctx);
/* A second prover over the SAME key on the donor's device that borrows the donor's point planes
 * (no second copy of the 21 GiB at 2^22) and owns everything else (streams, sort state, workspaces):
 * two host threads can then keep two proofs in flight, one per ctx -- the front of one (digit sort,
 * witness map) runs under the bucket reductions and the finalisation of the other.  key / a / b as
 * given to the donor (the descriptor's query pointers are not read again).  The planes are reference
 * counted: g16_ctx_destroy(donor) while siblings live only retires the donor's HANDLE (it must not be
 * used again) -- its device state is freed by the destroy of the last sibling.  Throughput mode of a
 * proving service; one proof's latency does not change.                                            */
g16_status g16_ctx_create_sibling(g16_ctx* donor, const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                                  uint32_t num_constraints, const g16_options* opt, g16_ctx** out);
const char* g16_last_error(const g16_ctx* ctx); /* ctx may be NULL: error of the last failed create */

/* Single-process multi-device prover (`device_ids, n_dev`; section 8(e)).
 * One sharded rank per listed device INSIDE the library: the MSMs are cut over the devices as
 * opt->shard says (point ranges by default, G16_SHARD_* below) and (n_dev a power of two) the witness
 * map becomes four-step NTTs whose two
 * all-to-all transposes are hipMemcpyPeerAsync pushes over xGMI, one copy stream per peer link;
 * the 1 KiB partial records are peer-copied to device_ids[0] and summed there.  The returned ctx is
 * used like a single-device one: g16_prove / g16_prove_dev (w_dev on device_ids[0]) shard
 * transparently, nothing of a proof touches the host between the witness upload and the 256-byte
 * download.  Replaces the same reference call as g16_ctx_create + g16_prove
 * (benches/groth16.rs:52-60); a Rust caller needs no launcher and no collective library.
 * opt->device / rank / world / dist_wm are ignored (dist_wm < 0 forces a replicated witness map).
 * device_ids may repeat an ordinal (several ranks time-sharing one GPU: functional tests).
 * Direct peer access between the devices is requested and REPORTED (g16_ctx_info out[14]; one line on
 * stderr when the runtime will stage copies; G16_REQUIRE_PEER_ACCESS=1 makes that an error).
 * Creation ends with a self-test of every peer path (a 4 KiB-per-pair all-to-all echo and a gather of
 * the partial records through the copy streams, events and buffers a proof uses, known patterns
 * checked on the devices): a path that does not deliver what was sent is an ERROR here (G16_ERR_HIP,
 * g16_last_error(NULL) names the exchange and the source / destination ranks and devices), never a
 * wrong proof later.                                                                                */
g16_status g16_ctx_create_multi(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                                uint32_t num_constraints, const int* device_ids, int n_dev,
                                const g16_options* opt, g16_ctx** out);

/* CircomReduction::witness_map_from_matrices (src/circom/qap.rs:23-88).
 * w: full_assignment, n_vars x 4 u64; h_out: domain_size x 4 u64 (natural order, Montgomery).   */
g16_status g16_witness_map(g16_ctx* ctx, const uint64_t* w, size_t n_vars, uint64_t* h_out);

/* VariableBaseMSM::msm_bigint over one resident query (ark-ec; reached from
 * create_proof_with_assignment).  scalars: len x 4 u64 Montgomery Fr; pairs scalar i with
 *   A/B1: query[1 + i]   (assignment = w[1..], as `msm(&query[1..], assignment)` upstream)
 *   L   : l_query[i]     (aux assignment = w[num_inputs..])
 *   H   : h_query[i]
 * out: affine point (64 bytes).  Only valid on a world == 1 ctx.                                  */
g16_status g16_msm_g1(g16_ctx* ctx, int which, const uint64_t* scalars, size_t len, uint8_t out[64]);
/* Same for b_g2_query[1 + i]; out: 128 bytes.                                                     */
g16_status g16_msm_g2(g16_ctx* ctx, const uint64_t* scalars, size_t len, uint8_t out[128]);
/* Device-resident variants (scalars / witness / h in HBM; BASELINE config 2 times the witness map
 * and each MSM separately without PCIe in the way).  h_dev_out: domain_size x 32 bytes, Montgomery. */
g16_status g16_witness_map_dev(g16_ctx* ctx, const void* w_dev, size_t n_vars, void* h_dev_out);
g16_status g16_msm_g1_dev(g16_ctx* ctx, int which, const void* scalars_dev, size_t len, uint8_t out[64]);
g16_status g16_msm_g2_dev(g16_ctx* ctx, const void* scalars_dev, size_t len, uint8_t out[128]);

/* Groth16::<Bn254,CircomReduction>::create_proof_with_reduction_and_matrices
 * (benches/groth16.rs:52-60, src/zkey.rs:903-911) with pk/matrices/num_inputs/num_constraints
 * taken from the ctx.  r, s: 4 u64 Montgomery Fr.  proof_out: A|B|C affine.  world == 1 only.     */
g16_status g16_prove(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4], const uint64_t* w,
                     size_t n_vars, uint8_t proof_out[G16_PROOF_BYTES]);
/* Same with the witness already resident in HBM (device pointer, n_vars x 32 bytes).             */
g16_status g16_prove_dev(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4], const void* w_dev,
                         size_t n_vars, uint8_t proof_out[G16_PROOF_BYTES]);

/* Multi-GPU (one process per GPU): every rank computes the sums of ITS point range -- and, while
 * its remaining MSMs run, the two products s*A_rank and r*B1_rank the finalisation is linear in --
 * the host framework all-gathers the G16_PARTIAL_BYTES records (RCCL all_gather; EC addition is
 * not an ncclRedOp, so "all-reduce" = all-gather + local add), then any rank finishes the proof
 * with fixed-base table sums only.  partials: world x G16_PARTIAL_BYTES in rank order.             */
g16_status g16_prove_partial(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4],
                             const uint64_t* w, size_t n_vars,
                             uint8_t partial_out[G16_PARTIAL_BYTES]);
g16_status g16_prove_partial_dev(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4],
                                 const void* w_dev, size_t n_vars,
                                 uint8_t partial_out[G16_PARTIAL_BYTES]);
g16_status g16_prove_finish(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4],
                            const uint8_t* partials, int world, uint8_t proof_out[G16_PROOF_BYTES]);

/* Device-side hand-offs for host frameworks that own a stream (one process per GPU, torch / RCCL):
 * after g16_dist_set_exchange_stream(ctx, hipStream_t, 1) the phase calls below never block the
 * host -- the registered stream is made to wait (hipStreamWaitEvent) for each send buffer / partial
 * record, and every phase waits for what the caller has enqueued on that stream so far (its
 * all-to-all / all-gather).  g16_partial_buffer(): this rank's record in HBM (G16_PARTIAL_BYTES),
 * the all-gather's input; g16_gather_buffer(): world x G16_PARTIAL_BYTES, its output;
 * g16_prove_finish_dev() consumes the latter in place.  Without a registered stream the calls
 * block until their output is complete (frameworks that cannot share a stream).                    */
g16_status g16_dist_set_exchange_stream(g16_ctx* ctx, void* hip_stream, int enabled);
void* g16_partial_buffer(g16_ctx* ctx);
void* g16_gather_buffer(g16_ctx* ctx);
g16_status g16_prove_finish_dev(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4],
                                uint8_t proof_out[G16_PROOF_BYTES]);

/* Fully sharded prover (ctx created with options.dist_wm = 1, world a power of two): the witness map
 * (CircomReduction::witness_map_from_matrices, src/circom/qap.rs:23-88) is split over the ranks as
 * four-step NTTs whose two transposes are all-to-all exchanges the host framework performs
 * (RCCL all_to_all over xGMI) between the three phases.  send/recv: device buffers of
 * g16_dist_exchange_bytes() bytes, `world` equal chunks in rank order (all_to_all_single layout).
 *   phase1(r, s, w_dev, send)  -> exchange 1 -> phase2(recv, send) -> exchange 2
 *   -> phase3(recv, partial_out) -> all-gather of the partial records -> g16_prove_finish.
 * The rank's A / B1 / L / B2 MSMs run on the ctx's main stream during the exchanges.               */
size_t g16_dist_exchange_bytes(const g16_ctx* ctx);
g16_status g16_prove_dist_phase1(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4],
                                 const void* w_dev, size_t n_vars, void* send_dev);
g16_status g16_prove_dist_phase2(g16_ctx* ctx, const void* recv_dev, void* send_dev);
/* partial_out may be NULL when an exchange stream is registered: the record then stays in
 * g16_partial_buffer() and the registered stream waits for it.                                     */
g16_status g16_prove_dist_phase3(g16_ctx* ctx, const void* recv_dev,
                                 uint8_t partial_out[G16_PARTIAL_BYTES]);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------ */
#define G16_N_STAGES 10
g16_status g16_set_profiling(g16_ctx* ctx, int enabled);
/* HIP-event times accumulated since the last call; resets the accumulators.  Stages, in order:
 * witness_map, msm_sort, msm_accumulate_g1 (L, H), msm_accumulate_g2 (B2), msm_reduce, finalize,
 * msm_accumulate_g1_pair (A | B1 in one launch), msm_fixup (the exact additions behind an optimistic
 * G1 launch); g16_stage_name() returns the same strings.                                            */
g16_status g16_stage_times(g16_ctx* ctx, float ms[G16_N_STAGES], uint32_t launches[G16_N_STAGES]);
const char* g16_stage_name(int stage);
/* sizes chosen at create time: out[0]=c_w out[1]=W_w out[2]=planes_w out[3]=D_w, [4..7] same for H,
 * out[8] = domain_size, out[9] = log2(domain_size), out[10] / out[11] = points of the witness / H
 * shard (rank 0's for a multi-device ctx), out[12] = devices, out[13] = how a world > 1 ctx shards
 * (G16_SHARD_POINTS / G16_SHARD_BUCKETS; 0 for world = 1), out[14] = multi-device ctx: 1 when every
 * pair of its devices has direct peer access, 2 when some exchanges are staged by the runtime,
 * out[15] = bit 0: g16_prove goes through the fixed-base tables (g16_options.fixed_tables); bit 1: the
 * B2 (G2) MSM runs over a filtered view of the witness sort (>= 1/8 of b_g1_query / b_g2_query is the
 * point at infinity: wires that appear in no B row of a real circom circuit)                        */
g16_status g16_ctx_info(const g16_ctx* ctx, uint32_t out[16]);
/* Multi-device ctx (g16_ctx_create_multi with a distributed witness map): what every ordered (source,
 * destination) pair of its ranks delivered at create time, measured with the copies a proof makes --
 * gbps[src * n + dst] = GB/s of one peer copy of *probe_bytes (<= 64 MiB) from src's exchange buffer
 * into dst's; echo_us[src * n + dst] = microseconds of a 4 KiB copy there and back (host-timed, launch
 * latency included).  n = g16_ctx_info out[12]; at most `cap` entries of each table are written.  The
 * reference has nothing to compare (single process, CPU); this is the per-link figure the scaling
 * projection assumes, turned into a measurement on the first multi-GPU box.      */
g16_status g16_multi_links(const g16_ctx* ctx, float* gbps, float* echo_us, int cap, uint64_t* probe_bytes);
/* device pointer of the ctx's witness staging buffer (n_vars x 32 bytes) for g16_prove_dev        */
void* g16_witness_buffer(g16_ctx* ctx);
/* Makes the witness resident: copies it into the ctx's device staging buffer -- into EVERY device's
 * for a multi-device ctx -- and returns when it is there.  g16_prove_dev(ctx, r, s,
 * g16_witness_buffer(ctx), ...) then proves from the resident copies without moving the witness
 * (the multi-device ctx otherwise peer-broadcasts a device-resident witness from the first device
 * inside every call).                                                                             */
g16_status g16_witness_upload(g16_ctx* ctx, const uint64_t* w, size_t n_vars);
/* page-locked HOST staging buffer (n_vars x 32 bytes, owned by the ctx): a caller that writes the
 * full assignment here (instead of into a Vec) gets the H2D copy of g16_prove at PCIe line rate   */
void* g16_witness_host_buffer(g16_ctx* ctx);

/* Constraint satisfaction of a witness: (A_i . w)(B_i . w) == C_i . w for every row, the check
 * CircomBuilder::build performs in debug builds (reference src/circom/builder.rs:101-114; the
 * reference's unit test at src/circom/circuit.rs:92-107 asserts cs.is_satisfied()).  a, b, c: the
 * R1CS rows as CSR (g16_r1cs_matrices); *first_unsatisfied = row index, or -1 when satisfied.     */
g16_status g16_check_satisfied(int device, const g16_csr* a, const g16_csr* b, const g16_csr* c,
                               uint32_t num_constraints, const uint64_t* w, size_t n_vars,
                               int64_t* first_unsatisfied);

/* ---- batch verification (not on the proving path) --------------- */
/* VerifyingKey<Bn254> in the packed forms read_zkey produces (src/zkey.rs:241-257): ic =
 * gamma_abc_g1, ic_count = n_public + 1 points of 64 bytes.                                        */
typedef struct {
  uint8_t alpha_g1[64];
  uint8_t beta_g2[128], gamma_g2[128], delta_g2[128];
  const uint8_t* ic;
  uint32_t ic_count;
} g16_vk_desc;
/* Groth16::process_vk + verify_with_processed_vk (reference call sites src/zkey.rs:868-870,914-916;
 * tests/groth16.rs:33-35) for n_proofs proofs under one key, one GPU lane per proof:
 *   ok_out[i] = 1 iff e(A_i, B_i) = e(alpha, beta) e(IC_0 + sum_j pub_ij IC_{j+1}, gamma) e(C_i, delta)
 * proofs: n_proofs x G16_PROOF_BYTES (A | B | C as g16_prove writes them); public_inputs:
 * n_proofs x (ic_count - 1) x 4 u64 Montgomery Fr.  The reference only ever pairs a DESERIALISED
 * Proof, and ark-serialize (Validate::Yes) rejects what this call therefore rejects itself with
 * ok = 0 before any pairing: a coordinate that is not canonical (stored value >= q), a point off its
 * curve, a B outside the prime-order subgroup of the twist ([r] B != infinity).                     */
g16_status g16_verify_batch(int device, const g16_vk_desc* vk, const uint8_t* proofs,
                            const uint64_t* public_inputs, uint32_t n_proofs, uint8_t* ok_out);

/* ---- RCCL inside the library (north_star: "a final RCCL all-reduce of partial bucket sums over xGMI") ---- */
/* A host that is not PyTorch (the Rust shim) creates one per-rank ctx per process (g16_options.rank / world,
 * dist_wm = 1) and ONE ncclComm_t over the same ranks with its own RCCL (ncclGetUniqueId / ncclCommInitRank),
 * then hands it over: nccl_comm is that opaque handle.  The library resolves ncclAllToAll / ncclAllGather /
 * ncclCommCount from the RCCL already loaded in the process (dlsym(RTLD_DEFAULT)), else from librccl.so -- it does
 * not link RCCL.  The collectives MUST run in the same RCCL copy that created the communicator: a host that links
 * RCCL has it in the global symbol scope; a host that dlopen()s it (Python's ctypes) must open it with RTLD_GLOBAL
 * (tests/test_gpu_large.py, scripts/rccl_inlib_ranks.py do).  The library checks that the communicator has g16_options.world ranks, allocates the two exchange buffers and a
 * high-priority exchange stream.  Afterwards g16_prove_dist is one whole sharded proof per rank:
 *   phase 1 -> ncclAllToAll -> phase 2 -> ncclAllToAll -> phase 3 -> ncclAllGather of the 1 KiB records -> finish
 * (the g16_prove_dist_phase* calls with the collectives in between, ordered by events: the host never blocks
 * between phases).  Every rank returns the same 256 proof bytes.  EC addition is not an ncclRedOp: the
 * "all-reduce of bucket sums" is this all-gather + a local sum of world records.
 * g16_dist_rccl_ranks: ranks of the attached communicator, 0 when none is attached.                     */
g16_status g16_dist_attach_rccl(g16_ctx* ctx, void* nccl_comm);
int g16_dist_rccl_ranks(const g16_ctx* ctx);
g16_status g16_prove_dist(g16_ctx* ctx, const uint64_t r[4], const uint64_t s[4], const void* w_dev,
                          size_t n_vars, uint8_t proof_out[G16_PROOF_BYTES]);

/* ---- EvaluationDomain::fft_in_place / ifft_in_place ----------------------------- */
/* ark-poly Radix2EvaluationDomain::{fft_in_place, ifft_in_place} as called from reference
 * src/circom/qap.rs:60-61,72-73,79-81: in-place size-2^log_n transform of host data (Montgomery Fr),
 * natural order in and out, 1/n folded into the inverse.  impl selects which of the library's two
 * transform stacks runs it: 2 = the witness map's lazy-limb DIF kernels + bit reversal (ntt29.hip; the
 * default a caller wants), 3 = bit reversal + its DIT kernels (forward only); 0 / 1 = the same two
 * schedules on the saturated-limb kernels of ntt.hip, which the key generator uses.                */
g16_status g16_fft_in_place(int device, uint64_t* data, int log_n, int inverse, int impl);

#ifdef G16_DEBUG_ABI
/* Measurement only, NOT part of the product library: compiled when the library is built with
 * EXTRA=-DG16_DEBUG_ABI (scripts/alu_bench.py).  Integer-ALU ceilings: kind 0 = Fq Montgomery
 * multiplications, 1 = raw v_mad_u64_u32, 2 = G1 mixed additions (saturated limbs), 3 / 4 = G1 / G2 mixed
 * additions on the lazy limbs the MSM kernels use.  Returns elapsed seconds and the number of operations. */
g16_status g16_debug_alu_bench(int device, int kind, uint32_t blocks, uint32_t iters,
                               double* seconds, double* ops);
#endif

/* ---- synthetic keys (not on the proving path) --------------------- */
/* Trapdoor (known toxic waste) circom/snarkjs-style setup on the GPU: what
 * Groth16::generate_random_parameters_with_reduction::<CircomReduction> computes (call shape:
 * reference tests/groth16.rs:25; H basis: CircomReduction::h_query_scalars, src/circom/qap.rs:90-105).
 * at/bt/ct: TRANSPOSED constraint matrices (row = wire, col = constraint, Montgomery coefficients),
 * `at` including the n_public+1 rows snarkjs appends (row m+i holds coefficient 1 on signal i).
 * toxic: tau, alpha, beta, gamma, delta as 5 x 4 u64 Montgomery Fr.                               */
typedef struct g16_setup g16_setup;
g16_status g16_setup_create(int device, const g16_csr* at, const g16_csr* bt, const g16_csr* ct,
                            uint32_t n_vars, uint32_t n_public, uint32_t num_constraints,
                            const uint64_t* toxic, g16_setup** out);
/* Same with the reduction named (G16_REDUCTION_*): the H query is CircomReduction's or
 * LibsnarkReduction's h_query_scalars (the rest of the key is identical: CircomReduction
 * delegates instance_map_with_evaluation to LibsnarkReduction, src/circom/qap.rs:16-21).          */
g16_status g16_setup_create_ex(int device, const g16_csr* at, const g16_csr* bt, const g16_csr* ct,
                               uint32_t n_vars, uint32_t n_public, uint32_t num_constraints,
                               const uint64_t* toxic, int reduction, g16_setup** out);
/* host arrays owned by the handle; ic: (n_public+1) x 64 bytes = vk.gamma_abc_g1                  */
g16_status g16_setup_key(g16_setup* s, g16_key_desc* key, const uint8_t** ic, uint32_t* ic_count,
                         uint8_t gamma_g2[128]);
void g16_setup_destroy(g16_setup* s);

/* ---- loaders (host side, C++): see g16_loaders.h ---------------------------------------------- */

#ifdef __cplusplus
}
#endif
#endif /* G16_AMD_H */
