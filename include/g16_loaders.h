/* g16_loaders.h -- host-side (C++) loaders for the circom/snarkjs binary formats, exported over
 * the same C ABI.  They mirror the reference's loaders value-for-value and error-for-error:
 *   g16_zkey_*  <- read_zkey / BinFile (reference src/zkey.rs:53-60,73-133,151-196,288-368)
 *   g16_r1cs_*  <- R1CSFile::new + R1CS::from (reference src/circom/r1cs_reader.rs:26-39,54-249)
 *   g16_wtns_*  <- snarkjs .wtns (not parsed by the reference)
 * but hand back GPU-ready packed arrays (the on-disk point encoding IS the device encoding) from
 * one bulk read instead of one Read call per 32-byte field (src/zkey.rs:328-368).
 * All returned pointers are owned by the handle and stay valid until *_close.                    */
#ifndef G16_LOADERS_H
#define G16_LOADERS_H

#include "g16_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* g16_loader_last_error(void); /* thread-local message of the last failed loader call */

/* ---------------------------------------------------------------- .zkey ---------------------- */
typedef struct g16_zkey g16_zkey;

typedef struct {                 /* HeaderGroth (src/zkey.rs:270-317) */
  uint32_t n8q, n8r;
  uint8_t q[32], r[32];          /* little-endian */
  uint32_t n_vars, n_public, domain_size, power;
  uint8_t alpha_g1[64], beta_g1[64], beta_g2[128], gamma_g2[128], delta_g1[64], delta_g2[128];
} g16_zkey_header;

typedef struct {                 /* ConstraintMatrices as built by BinFile::matrices (src/zkey.rs:151-196) */
  uint32_t num_instance_variables; /* n_public + 1              (:182) */
  uint32_t num_witness_variables;  /* n_vars - n_public         (:183) */
  uint32_t num_constraints;        /* max constraint - n_public (:171) */
  uint64_t a_num_non_zero, b_num_non_zero;
  g16_csr a, b;                    /* coefficients Montgomery (file value / R, :322-325) */
} g16_matrices;

/* g16_zkey_open maps the file (the key arrays handed to g16_ctx_create are views of the page cache):
 * the file must stay unchanged until g16_zkey_close -- a zkey truncated or rewritten underneath an
 * open handle faults (SIGBUS) instead of returning G16_ERR_IO.  G16_ZKEY_COPY=1 in the environment
 * reads it into owned memory instead (the behaviour of g16_zkey_open_mem).                        */
g16_status g16_zkey_open(const char* path, g16_zkey** out);
g16_status g16_zkey_open_mem(const uint8_t* data, size_t len, g16_zkey** out); /* data is copied */
void g16_zkey_close(g16_zkey* z);
g16_status g16_zkey_header_get(const g16_zkey* z, g16_zkey_header* out);
/* ProvingKey arrays (zero-copy views of sections 5..9) + header points */
g16_status g16_zkey_key(const g16_zkey* z, g16_key_desc* out);
/* vk.gamma_abc_g1 = IC, (n_public + 1) x 64 bytes (section 3) */
const uint8_t* g16_zkey_ic(const g16_zkey* z, uint32_t* count);
g16_status g16_zkey_matrices(g16_zkey* z, g16_matrices* out);

/* snarkjs-format .zkey WRITER: the inverse of g16_zkey_open for a key held in packed arrays (e.g.
 * one minted by g16_setup_create) -- lets synthetic circuits go through the same file format and
 * loader path as circom/snarkjs artefacts (reference format notes: src/zkey.rs:1-27).  ic:
 * (n_public + 1) x 64 bytes; a, b: ConstraintMatrices rows WITHOUT the n_public + 1 rows snarkjs
 * appends (they are generated).                                                                    */
g16_status g16_zkey_write(const char* path, const g16_key_desc* key, const uint8_t* ic,
                          const uint8_t gamma_g2[128], const g16_csr* a, const g16_csr* b,
                          uint32_t num_constraints);

/* ---------------------------------------------------------------- .r1cs ---------------------- */
typedef struct g16_r1cs g16_r1cs;

typedef struct {
  uint32_t version;
  uint32_t field_size;
  uint8_t prime[32];
  uint32_t n_wires, n_pub_out, n_pub_in, n_prv_in;
  uint64_t n_labels;
  uint32_t n_constraints;
  /* R1CS::from (r1cs_reader.rs:26-39) */
  uint32_t num_inputs, num_aux, num_variables;
} g16_r1cs_header;

g16_status g16_r1cs_open(const char* path, g16_r1cs** out);
g16_status g16_r1cs_open_mem(const uint8_t* data, size_t len, g16_r1cs** out);
void g16_r1cs_close(g16_r1cs* r);
g16_status g16_r1cs_header_get(const g16_r1cs* r, g16_r1cs_header* out);
/* the three linear-combination lists per constraint as CSR (column = wire index, coefficient
 * Montgomery, as F::deserialize_uncompressed leaves it in memory, r1cs_reader.rs:203-213) */
g16_status g16_r1cs_matrices(const g16_r1cs* r, g16_csr* a, g16_csr* b, g16_csr* c);
const uint64_t* g16_r1cs_wire_mapping(const g16_r1cs* r, uint32_t* count);

/* ---------------------------------------------------------------- .wtns ---------------------- */
/* Reads n witness values; *out (malloc'd, free with g16_free) holds n x 4 u64 Montgomery Fr.      */
g16_status g16_wtns_read(const char* path, uint64_t** out, uint32_t* n);
g16_status g16_wtns_read_mem(const uint8_t* data, size_t len, uint64_t** out, uint32_t* n);
void g16_free(void* p);

/* canonical little-endian 32-byte integers <-> Montgomery Fr (host helpers for callers that hold
 * decimal / canonical witnesses, e.g. snarkjs JSON)                                               */
g16_status g16_fr_from_canonical(const uint8_t* in, uint64_t* out, size_t n);
g16_status g16_fr_to_canonical(const uint64_t* in, uint8_t* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* G16_LOADERS_H */
