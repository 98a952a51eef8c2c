// loaders.cpp -- host-side loaders behind include/g16_loaders.h.
//
//   zkey : reference src/zkey.rs  (BinFile::new :73-101, proving_key :103-133, matrices :151-196,
//          HeaderGroth::read :288-317, deserialize_field_fr :322-325, deserialize_g1/g2 :340-360)
//   r1cs : reference src/circom/r1cs_reader.rs (R1CSFile::new :54-146, Header::new :161-200,
//          read_constraint_vec :203-213, read_constraints :215-229, read_map :231-249, R1CS::from :26-39)
// The point sections are handed out as zero-copy views: the on-disk encoding (x|y Montgomery LE,
// all-zero = infinity) is already the device encoding.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/g16_loaders.h"
#include "field.h"

using g16::Fr;
using g16::U256;

namespace {

thread_local std::string t_err;

g16_status fail(g16_status code, const std::string& m) {
  t_err = m;
  return code;
}

bool read_file(const char* path, std::vector<uint8_t>& out, std::string& err) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    err = std::string("cannot open ") + path;
    return false;
  }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz < 0) {
    fclose(f);
    err = "cannot stat file";
    return false;
  }
  out.resize((size_t)sz);
  size_t got = sz ? fread(out.data(), 1, (size_t)sz, f) : 0;
  fclose(f);
  if (got != (size_t)sz) {
    err = "short read";
    return false;
  }
  return true;
}

struct Cursor {
  const uint8_t* p;
  size_t n, o = 0;
  bool ok = true;
  Cursor(const uint8_t* p_, size_t n_, size_t o_ = 0) : p(p_), n(n_), o(o_) {}
  bool need(size_t k) {
    if (!ok || o + k > n || o + k < o) {
      ok = false;
      return false;
    }
    return true;
  }
  uint32_t u32() {
    if (!need(4)) return 0;
    uint32_t v;
    memcpy(&v, p + o, 4);
    o += 4;
    return v;
  }
  uint64_t u64() {
    if (!need(8)) return 0;
    uint64_t v;
    memcpy(&v, p + o, 8);
    o += 8;
    return v;
  }
  const uint8_t* bytes(size_t k) {
    if (!need(k)) return nullptr;
    const uint8_t* r = p + o;
    o += k;
    return r;
  }
  void skip(uint64_t k) {
    if (!ok || k > n - o) {
      ok = false;
      return;
    }
    o += (size_t)k;
  }
};

struct Section {
  size_t pos;
  uint64_t size;
};

const uint8_t kR1csPrime[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9,
                                0x79, 0x48, 0xe8, 0x33, 0x28, 0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45,
                                0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

bool fr_lt_modulus(const uint8_t* le32) {
  for (int i = 31; i >= 0; --i) {
    if (le32[i] < kR1csPrime[i]) return true;
    if (le32[i] > kR1csPrime[i]) return false;
  }
  return false;  // equal
}

Fr fr_load(const uint8_t* le32) {
  Fr a;
  memcpy(a.v, le32, 32);
  return a;
}

struct CsrOwned {
  std::vector<uint32_t> rowptr, col;
  std::vector<Fr> val;
  g16_csr view() const {
    g16_csr c;
    c.row_ptr = rowptr.data();
    c.col = col.data();
    c.coeff = (const uint64_t*)val.data();
    c.nnz = col.size();
    return c;
  }
};

}  // namespace

// Read-only view of a whole file.  g16_zkey_open maps the file (the 64 / 128-byte point sections are
// handed to g16_ctx_create straight out of the page cache: no 6 GiB host copy at 2^24, and the
// reference's one-Read-call-per-field-element loop, src/zkey.rs:328-368, becomes page faults);
// g16_zkey_open_mem owns a copy of the caller's buffer.
struct FileView {
  const uint8_t* p = nullptr;
  size_t n = 0;
  void* map = nullptr;  // non-null: munmap on release
  std::vector<uint8_t> own;
  FileView() = default;
  FileView(const FileView&) = delete;
  FileView& operator=(const FileView&) = delete;
  ~FileView() {
    if (map) munmap(map, n);
  }
  void copy_of(const uint8_t* d, size_t len) {
    own.assign(d, d + len);
    p = own.data();
    n = own.size();
  }
  bool map_file(const char* path, std::string& err) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
      err = std::string("cannot open ") + path;
      return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) {
      close(fd);
      err = "cannot stat file";
      return false;
    }
    n = (size_t)st.st_size;
    if (n == 0) {  // mmap of length 0 is an error; an empty file is a truncated header downstream
      close(fd);
      p = own.data();
      return true;
    }
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
      err = "mmap failed";
      n = 0;
      return false;
    }
    // No MADV_SEQUENTIAL: the sharded provers read the H section strided (global_index(t)) and a
    // multi-device ctx uploads every section once per device.  G16_ZKEY_COPY=1 reads the file into
    // owned memory instead of mapping it: a zkey that is truncated or rewritten while the handle is
    // open then cannot raise SIGBUS inside a later upload (the mapping requires the file to stay
    // unchanged until g16_zkey_close; include/g16_loaders.h says so).
    if (const char* e = getenv("G16_ZKEY_COPY")) {
      if (e[0] == '1') {
        own.assign((const uint8_t*)m, (const uint8_t*)m + n);
        munmap(m, n);
        p = own.data();
        return true;
      }
    }
    // sequential access hint only: MADV_WILLNEED on a multi-10-GB zkey would start a whole-file
    // readahead up front, while the sections are uploaded one after another
    (void)madvise(m, n, MADV_SEQUENTIAL);
    map = m;
    p = (const uint8_t*)m;
    return true;
  }
  const uint8_t* data() const { return p; }
  size_t size() const { return n; }
};

struct g16_zkey {
  FileView data;
  std::map<uint32_t, Section> sec;  // first occurrence wins (get_section, zkey.rs:135-137)
  g16_zkey_header hdr;
  bool have_matrices = false;
  g16_matrices mat;
  CsrOwned A, B;
};

struct g16_r1cs {
  g16_r1cs_header hdr;
  CsrOwned A, B, C;
  std::vector<uint64_t> wire_mapping;
};

namespace {

g16_status zkey_parse(g16_zkey* z) {
  const uint8_t* d = z->data.data();
  const size_t n = z->data.size();
  Cursor c(d, n);
  c.bytes(4);  // magic: the reference does not check it (zkey.rs:74-75)
  (void)c.u32();
  const uint32_t nsec = c.u32();
  if (!c.ok) return fail(G16_ERR_IO, "zkey: truncated file header");
  for (uint32_t i = 0; i < nsec; ++i) {
    const uint32_t id = c.u32();
    const uint64_t len = c.u64();
    if (!c.ok) return fail(G16_ERR_IO, "zkey: truncated section table");
    if (!z->sec.count(id)) z->sec[id] = Section{c.o, len};
    c.skip(len);
    if (!c.ok) return fail(G16_ERR_IO, "zkey: section runs past the end of the file");
  }
  for (uint32_t id : {2u, 3u, 4u, 5u, 6u, 7u, 8u, 9u})
    if (!z->sec.count(id)) return fail(G16_ERR_IO, "zkey: missing section " + std::to_string(id));
  // HeaderGroth::read (zkey.rs:288-317)
  Cursor h(d, n, z->sec[2].pos);
  g16_zkey_header& H = z->hdr;
  memset(&H, 0, sizeof H);
  H.n8q = h.u32();
  if (H.n8q != 32) return fail(G16_ERR_IO, "zkey: n8q != 32");
  const uint8_t* q = h.bytes(32);
  H.n8r = h.u32();
  if (!h.ok || H.n8r != 32) return fail(G16_ERR_IO, "zkey: n8r != 32");
  const uint8_t* r = h.bytes(32);
  H.n_vars = h.u32();
  H.n_public = h.u32();
  H.domain_size = h.u32();
  const uint8_t* pts = h.bytes(64 + 64 + 128 + 128 + 64 + 128);
  if (!h.ok) return fail(G16_ERR_IO, "zkey: truncated groth16 header");
  memcpy(H.q, q, 32);
  memcpy(H.r, r, 32);
  uint32_t pw = 0;
  while (((uint64_t)1 << pw) < H.domain_size) ++pw;  // ark_std::log2 = ceil(log2)
  H.power = pw;
  // ZVerifyingKey::new order (zkey.rs:241-257): alpha1, beta1, beta2, gamma2, delta1, delta2
  memcpy(H.alpha_g1, pts, 64);
  memcpy(H.beta_g1, pts + 64, 64);
  memcpy(H.beta_g2, pts + 128, 128);
  memcpy(H.gamma_g2, pts + 256, 128);
  memcpy(H.delta_g1, pts + 384, 64);
  memcpy(H.delta_g2, pts + 448, 128);
  // 64-bit comparisons: n_public = 0xFFFFFFFF must not wrap n_public + 1 to 0
  if ((uint64_t)H.n_public + 1 > (uint64_t)H.n_vars)
    return fail(G16_ERR_IO, "zkey: n_vars < n_public + 1");
  if ((uint64_t)H.n_public + 1 > (uint64_t)H.domain_size)
    return fail(G16_ERR_IO, "zkey: domain_size < n_public + 1");
  // section sizes the proving_key() reads rely on (zkey.rs:107-111)
  struct Need {
    uint32_t id;
    uint64_t bytes;
  } needs[] = {{3, ((uint64_t)H.n_public + 1) * 64},
               {5, (uint64_t)H.n_vars * 64},
               {6, (uint64_t)H.n_vars * 64},
               {7, (uint64_t)H.n_vars * 128},
               {8, ((uint64_t)H.n_vars - H.n_public - 1) * 64},
               {9, (uint64_t)H.domain_size * 64}};
  for (auto& nd : needs)
    if (z->sec[nd.id].size < nd.bytes)
      return fail(G16_ERR_IO, "zkey: section " + std::to_string(nd.id) + " is too short");
  return G16_OK;
}

g16_status r1cs_parse(const uint8_t* d, size_t n, g16_r1cs* R) {
  Cursor c(d, n);
  const uint8_t* magic = c.bytes(4);
  if (!magic || memcmp(magic, "r1cs", 4) != 0) return fail(G16_ERR_IO, "Invalid magic number");
  const uint32_t version = c.u32();
  if (!c.ok) return fail(G16_ERR_IO, "unexpected end of file");
  if (version != 1) return fail(G16_ERR_IO, "Unsupported version");
  const uint32_t nsec = c.u32();
  std::map<uint32_t, Section> sec;  // later duplicates overwrite (HashMap::insert, :85-86)
  for (uint32_t i = 0; i < nsec; ++i) {
    const uint32_t ty = c.u32();
    const uint64_t sz = c.u64();
    if (!c.ok) return fail(G16_ERR_IO, "unexpected end of file");
    sec[ty] = Section{c.o, sz};
    c.skip(sz);
    if (!c.ok) return fail(G16_ERR_IO, "unexpected end of file");
  }
  if (!sec.count(1)) return fail(G16_ERR_IO, "No section offset for header type found");
  g16_r1cs_header& H = R->hdr;
  memset(&H, 0, sizeof H);
  H.version = version;
  {
    Cursor h(d, n, sec[1].pos);
    H.field_size = h.u32();
    if (!h.ok) return fail(G16_ERR_IO, "unexpected end of file");
    if (H.field_size != 32) return fail(G16_ERR_IO, "This parser only supports 32-byte fields");
    if (sec[1].size != 32 + (uint64_t)H.field_size)
      return fail(G16_ERR_IO, "Invalid header section size");
    const uint8_t* prime = h.bytes(32);
    if (!prime) return fail(G16_ERR_IO, "unexpected end of file");
    if (memcmp(prime, kR1csPrime, 32) != 0) return fail(G16_ERR_IO, "This parser only supports bn256");
    memcpy(H.prime, prime, 32);
    H.n_wires = h.u32();
    H.n_pub_out = h.u32();
    H.n_pub_in = h.u32();
    H.n_prv_in = h.u32();
    H.n_labels = h.u64();
    H.n_constraints = h.u32();
    if (!h.ok) return fail(G16_ERR_IO, "unexpected end of file");
  }
  if (!sec.count(2)) return fail(G16_ERR_IO, "No section offset for constraint type found");
  {
    Cursor k(d, n, sec[2].pos);
    CsrOwned* M[3] = {&R->A, &R->B, &R->C};
    for (auto* m : M) {
      m->rowptr.assign(1, 0);
      m->rowptr.reserve((size_t)H.n_constraints + 1);
    }
    for (uint32_t i = 0; i < H.n_constraints; ++i) {
      for (auto* m : M) {
        const uint32_t cnt = k.u32();
        if (!k.ok) return fail(G16_ERR_IO, "unexpected end of file");
        for (uint32_t j = 0; j < cnt; ++j) {
          const uint32_t wire = k.u32();
          const uint8_t* v = k.bytes(32);
          if (!v) return fail(G16_ERR_IO, "unexpected end of file");
          // F::deserialize_uncompressed rejects non-canonical values
          if (!fr_lt_modulus(v)) return fail(G16_ERR_IO, "invalid field element (>= modulus)");
          U256 u;
          memcpy(u.v, v, 32);
          m->col.push_back(wire);
          m->val.push_back(Fr::from_canonical(u));
        }
        m->rowptr.push_back((uint32_t)m->col.size());
      }
    }
  }
  if (!sec.count(3)) return fail(G16_ERR_IO, "No section offset for wire2label type found");
  {
    if (sec[3].size != (uint64_t)H.n_wires * 8) return fail(G16_ERR_IO, "Invalid map section size");
    Cursor w(d, n, sec[3].pos);
    R->wire_mapping.resize(H.n_wires);
    for (uint32_t i = 0; i < H.n_wires; ++i) R->wire_mapping[i] = w.u64();
    if (!w.ok) return fail(G16_ERR_IO, "unexpected end of file");
    if (H.n_wires == 0 || R->wire_mapping[0] != 0)
      return fail(G16_ERR_IO, "Wire 0 should always be mapped to 0");
  }
  H.num_inputs = 1 + H.n_pub_in + H.n_pub_out;
  H.num_variables = H.n_wires;
  H.num_aux = H.num_variables - H.num_inputs;
  return G16_OK;
}

g16_status wtns_parse(const uint8_t* d, size_t n, uint64_t** out, uint32_t* cnt) {
  Cursor c(d, n);
  const uint8_t* magic = c.bytes(4);
  if (!magic || memcmp(magic, "wtns", 4) != 0) return fail(G16_ERR_IO, "wtns: bad magic");
  (void)c.u32();
  const uint32_t nsec = c.u32();
  std::map<uint32_t, Section> sec;
  for (uint32_t i = 0; i < nsec; ++i) {
    const uint32_t ty = c.u32();
    const uint64_t sz = c.u64();
    if (!c.ok) return fail(G16_ERR_IO, "wtns: truncated");
    sec[ty] = Section{c.o, sz};
    c.skip(sz);
    if (!c.ok) return fail(G16_ERR_IO, "wtns: truncated");
  }
  if (!sec.count(1) || !sec.count(2)) return fail(G16_ERR_IO, "wtns: missing section");
  Cursor h(d, n, sec[1].pos);
  const uint32_t n8 = h.u32();
  if (!h.ok || n8 != 32) return fail(G16_ERR_IO, "wtns: only 32-byte fields are supported");
  const uint8_t* prime = h.bytes(32);
  if (!prime || memcmp(prime, kR1csPrime, 32) != 0) return fail(G16_ERR_IO, "wtns: not bn128");
  const uint32_t nw = h.u32();
  if (!h.ok || sec[2].size < (uint64_t)nw * 32) return fail(G16_ERR_IO, "wtns: truncated values");
  Cursor v(d, n, sec[2].pos);
  uint64_t* buf = (uint64_t*)malloc((size_t)(nw ? nw : 1) * 32);
  if (!buf) return fail(G16_ERR_INTERNAL, "out of memory");
  for (uint32_t i = 0; i < nw; ++i) {
    const uint8_t* b = v.bytes(32);
    if (!b || !fr_lt_modulus(b)) {
      free(buf);
      return fail(G16_ERR_IO, "wtns: invalid field element");
    }
    U256 u;
    memcpy(u.v, b, 32);
    Fr m = Fr::from_canonical(u);
    memcpy(buf + (size_t)i * 4, m.v, 32);
  }
  *out = buf;
  *cnt = nw;
  return G16_OK;
}

}  // namespace

extern "C" {

const char* g16_loader_last_error(void) { return t_err.c_str(); }

g16_status g16_zkey_open_mem(const uint8_t* data, size_t len, g16_zkey** out) {
  if (!data || !out) return fail(G16_ERR_INVALID, "null argument");
  *out = nullptr;
  g16_zkey* z = new g16_zkey();
  z->data.copy_of(data, len);
  g16_status st = zkey_parse(z);
  if (st != G16_OK) {
    delete z;
    return st;
  }
  *out = z;
  return G16_OK;
}

g16_status g16_zkey_open(const char* path, g16_zkey** out) {
  if (!path || !out) return fail(G16_ERR_INVALID, "null argument");
  *out = nullptr;
  g16_zkey* z = new g16_zkey();
  std::string err;
  if (!z->data.map_file(path, err)) {
    delete z;
    return fail(G16_ERR_IO, err);
  }
  g16_status st = zkey_parse(z);
  if (st != G16_OK) {
    delete z;
    return st;
  }
  *out = z;
  return G16_OK;
}

void g16_zkey_close(g16_zkey* z) { delete z; }

g16_status g16_zkey_header_get(const g16_zkey* z, g16_zkey_header* out) {
  if (!z || !out) return fail(G16_ERR_INVALID, "null argument");
  *out = z->hdr;
  return G16_OK;
}

g16_status g16_zkey_key(const g16_zkey* z, g16_key_desc* out) {
  if (!z || !out) return fail(G16_ERR_INVALID, "null argument");
  const uint8_t* d = z->data.data();
  memset(out, 0, sizeof *out);
  out->n_vars = z->hdr.n_vars;
  out->n_public = z->hdr.n_public;
  out->domain_size = z->hdr.domain_size;
  out->a_query = d + z->sec.at(5).pos;
  out->b_g1_query = d + z->sec.at(6).pos;
  out->b_g2_query = d + z->sec.at(7).pos;
  out->l_query = d + z->sec.at(8).pos;
  out->h_query = d + z->sec.at(9).pos;
  memcpy(out->alpha_g1, z->hdr.alpha_g1, 64);
  memcpy(out->beta_g1, z->hdr.beta_g1, 64);
  memcpy(out->delta_g1, z->hdr.delta_g1, 64);
  memcpy(out->beta_g2, z->hdr.beta_g2, 128);
  memcpy(out->delta_g2, z->hdr.delta_g2, 128);
  return G16_OK;
}

const uint8_t* g16_zkey_ic(const g16_zkey* z, uint32_t* count) {
  if (!z) return nullptr;
  if (count) *count = z->hdr.n_public + 1;
  return z->data.data() + z->sec.at(3).pos;
}

g16_status g16_zkey_matrices(g16_zkey* z, g16_matrices* out) {
  if (!z || !out) return fail(G16_ERR_INVALID, "null argument");
  if (!z->have_matrices) {
    // BinFile::matrices (zkey.rs:151-196)
    const uint8_t* d = z->data.data();
    const Section s4 = z->sec.at(4);
    Cursor c(d, s4.pos + (size_t)s4.size <= z->data.size() ? s4.pos + (size_t)s4.size : z->data.size(),
             s4.pos);
    const uint32_t ncoef = c.u32();
    if (!c.ok) return fail(G16_ERR_IO, "zkey: truncated coefficient section");
    const uint8_t* recs = c.bytes((size_t)ncoef * 44);
    if (!recs) return fail(G16_ERR_IO, "zkey: truncated coefficient section");
    const uint32_t dom = z->hdr.domain_size;
    uint32_t max_c = 0;
    std::vector<uint32_t> cnt[2];
    cnt[0].assign((size_t)dom + 1, 0);
    cnt[1].assign((size_t)dom + 1, 0);
    for (uint32_t i = 0; i < ncoef; ++i) {
      uint32_t m, row;
      memcpy(&m, recs + (size_t)i * 44, 4);
      memcpy(&row, recs + (size_t)i * 44 + 4, 4);
      if (m > 1) return fail(G16_ERR_IO, "zkey: coefficient with matrix index > 1");
      if (row >= dom) return fail(G16_ERR_IO, "zkey: coefficient row outside the domain");
      if (row > max_c) max_c = row;
      cnt[m][row + 1]++;
    }
    if (max_c < z->hdr.n_public) return fail(G16_ERR_IO, "zkey: fewer rows than public inputs");
    const uint32_t nc = max_c - z->hdr.n_public;  // :171
    CsrOwned* M[2] = {&z->A, &z->B};
    std::vector<uint32_t> cur[2];
    for (int k = 0; k < 2; ++k) {
      // rows >= nc are dropped (truncate, :173-175): arkworks re-adds the public-input rows
      M[k]->rowptr.assign((size_t)nc + 1, 0);
      for (uint32_t r = 0; r < nc; ++r) M[k]->rowptr[r + 1] = M[k]->rowptr[r] + cnt[k][r + 1];
      M[k]->col.resize(M[k]->rowptr[nc]);
      M[k]->val.resize(M[k]->rowptr[nc]);
      cur[k].assign(M[k]->rowptr.begin(), M[k]->rowptr.end());
    }
    for (uint32_t i = 0; i < ncoef; ++i) {  // stable: file order inside a row, like the push()
      const uint8_t* rec = recs + (size_t)i * 44;
      uint32_t m, row, sig;
      memcpy(&m, rec, 4);
      memcpy(&row, rec + 4, 4);
      memcpy(&sig, rec + 8, 4);
      if (row >= nc) continue;
      // the reference indexes full_assignment[signal] (evaluate_constraint) and panics out of
      // bounds; the kernels would read past the witness, so a bad wire index is a load error here
      if (sig >= z->hdr.n_vars) return fail(G16_ERR_IO, "zkey: coefficient signal index >= n_vars");
      const uint32_t at = cur[m][row]++;
      M[m]->col[at] = sig;
      // deserialize_field_fr (:322-325): the stored value is v*R^2; one Montgomery reduction
      // leaves v*R, the in-memory Montgomery form of v
      U256 u = fr_load(rec + 12).to_canonical();
      Fr v;
      memcpy(v.v, u.v, 32);
      M[m]->val[at] = v;
    }
    g16_matrices& mt = z->mat;
    mt.num_instance_variables = z->hdr.n_public + 1;              // :182
    mt.num_witness_variables = z->hdr.n_vars - z->hdr.n_public;   // :183
    mt.num_constraints = nc;
    mt.a_num_non_zero = z->A.col.size();
    mt.b_num_non_zero = z->B.col.size();
    mt.a = z->A.view();
    mt.b = z->B.view();
    z->have_matrices = true;
  }
  *out = z->mat;
  return G16_OK;
}

g16_status g16_r1cs_open_mem(const uint8_t* data, size_t len, g16_r1cs** out) {
  if (!data || !out) return fail(G16_ERR_INVALID, "null argument");
  *out = nullptr;
  g16_r1cs* r = new g16_r1cs();
  g16_status st = r1cs_parse(data, len, r);
  if (st != G16_OK) {
    delete r;
    return st;
  }
  *out = r;
  return G16_OK;
}

g16_status g16_r1cs_open(const char* path, g16_r1cs** out) {
  if (!path || !out) return fail(G16_ERR_INVALID, "null argument");
  std::vector<uint8_t> buf;
  std::string err;
  if (!read_file(path, buf, err)) return fail(G16_ERR_IO, err);
  return g16_r1cs_open_mem(buf.data(), buf.size(), out);
}

void g16_r1cs_close(g16_r1cs* r) { delete r; }

g16_status g16_r1cs_header_get(const g16_r1cs* r, g16_r1cs_header* out) {
  if (!r || !out) return fail(G16_ERR_INVALID, "null argument");
  *out = r->hdr;
  return G16_OK;
}

g16_status g16_r1cs_matrices(const g16_r1cs* r, g16_csr* a, g16_csr* b, g16_csr* c) {
  if (!r) return fail(G16_ERR_INVALID, "null argument");
  if (a) *a = r->A.view();
  if (b) *b = r->B.view();
  if (c) *c = r->C.view();
  return G16_OK;
}

const uint64_t* g16_r1cs_wire_mapping(const g16_r1cs* r, uint32_t* count) {
  if (!r) return nullptr;
  if (count) *count = (uint32_t)r->wire_mapping.size();
  return r->wire_mapping.data();
}

g16_status g16_wtns_read_mem(const uint8_t* data, size_t len, uint64_t** out, uint32_t* n) {
  if (!data || !out || !n) return fail(G16_ERR_INVALID, "null argument");
  return wtns_parse(data, len, out, n);
}

g16_status g16_wtns_read(const char* path, uint64_t** out, uint32_t* n) {
  if (!path || !out || !n) return fail(G16_ERR_INVALID, "null argument");
  std::vector<uint8_t> buf;
  std::string err;
  if (!read_file(path, buf, err)) return fail(G16_ERR_IO, err);
  return wtns_parse(buf.data(), buf.size(), out, n);
}

void g16_free(void* p) { free(p); }

/* snarkjs .zkey writer (format: SURVEY.md Appendix A.2 = what read_zkey parses, reference
 * src/zkey.rs:1-27,73-133,151-196,288-368).  Coefs(4) values are stored as v * R^2 (the reader
 * divides by R^2, zkey.rs:320-325), points in the Montgomery encoding the key arrays already have;
 * the n_public + 1 rows snarkjs appends (row m+i: A coefficient 1 on signal i) are written too.    */
g16_status g16_zkey_write(const char* path, const g16_key_desc* key, const uint8_t* ic,
                          const uint8_t gamma_g2[128], const g16_csr* a, const g16_csr* b,
                          uint32_t num_constraints) {
  if (!path || !key || !ic || !gamma_g2 || !a || !b) return fail(G16_ERR_INVALID, "null argument");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(G16_ERR_IO, std::string("cannot create ") + path);
  bool ok = true;
  auto put = [&](const void* p, size_t n) { ok = ok && (n == 0 || fwrite(p, 1, n, f) == n); };
  auto u32 = [&](uint32_t v) { put(&v, 4); };
  auto u64 = [&](uint64_t v) { put(&v, 8); };
  auto sec = [&](uint32_t id, uint64_t len) {
    u32(id);
    u64(len);
  };
  const uint32_t N = key->n_vars, p = key->n_public;
  const uint64_t ncoef = a->nnz + b->nnz + p + 1;
  put("zkey", 4);
  u32(1);
  u32(10);
  sec(1, 4);
  u32(1);  // groth16
  sec(2, 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128);
  u32(32);
  put(g16::FqParams::MOD, 32);
  u32(32);
  put(g16::FrParams::MOD, 32);
  u32(N);
  u32(p);
  u32(key->domain_size);
  put(key->alpha_g1, 64);
  put(key->beta_g1, 64);
  put(key->beta_g2, 128);
  put(gamma_g2, 128);
  put(key->delta_g1, 64);
  put(key->delta_g2, 128);
  sec(3, (uint64_t)(p + 1) * 64);
  put(ic, (size_t)(p + 1) * 64);
  sec(4, 4 + ncoef * (12 + 32));
  u32((uint32_t)ncoef);
  const Fr r2 = Fr::r2();
  auto coefs = [&](uint32_t which, const g16_csr* m) {
    for (uint32_t row = 0; row < num_constraints; ++row)
      for (uint32_t j = m->row_ptr[row]; j < m->row_ptr[row + 1]; ++j) {
        u32(which);
        u32(row);
        u32(m->col[j]);
        Fr v;
        memcpy(v.v, m->coeff + (size_t)j * 4, 32);
        // v holds the limbs of coeff * R; the Montgomery product with R^2 is (coeff R)(R^2)/R =
        // coeff * R^2 mod r: exactly the integer snarkjs stores
        const Fr vr = v * r2;
        put(vr.v, 32);
      }
  };
  coefs(0, a);
  coefs(1, b);
  for (uint32_t i = 0; i <= p; ++i) {
    u32(0);
    u32(num_constraints + i);
    u32(i);
    const Fr one_r2 = Fr::one() * r2;  // 1 * R^2
    put(one_r2.v, 32);
  }
  sec(5, (uint64_t)N * 64);
  put(key->a_query, (size_t)N * 64);
  sec(6, (uint64_t)N * 64);
  put(key->b_g1_query, (size_t)N * 64);
  sec(7, (uint64_t)N * 128);
  put(key->b_g2_query, (size_t)N * 128);
  sec(8, (uint64_t)(N - p - 1) * 64);
  put(key->l_query, (size_t)(N - p - 1) * 64);
  sec(9, (uint64_t)key->domain_size * 64);
  put(key->h_query, (size_t)key->domain_size * 64);
  sec(10, 4 + 64);
  u32(0);
  uint8_t zeros[64] = {0};
  put(zeros, 64);
  ok = (fclose(f) == 0) && ok;
  return ok ? G16_OK : fail(G16_ERR_IO, "short write");
}

g16_status g16_fr_from_canonical(const uint8_t* in, uint64_t* out, size_t n) {
  if (!in || !out) return fail(G16_ERR_INVALID, "null argument");
  for (size_t i = 0; i < n; ++i) {
    if (!fr_lt_modulus(in + i * 32)) return fail(G16_ERR_INVALID, "value >= field modulus");
    U256 u;
    memcpy(u.v, in + i * 32, 32);
    Fr m = Fr::from_canonical(u);
    memcpy(out + i * 4, m.v, 32);
  }
  return G16_OK;
}

g16_status g16_fr_to_canonical(const uint64_t* in, uint8_t* out, size_t n) {
  if (!in || !out) return fail(G16_ERR_INVALID, "null argument");
  for (size_t i = 0; i < n; ++i) {
    Fr m;
    memcpy(m.v, in + i * 4, 32);
    U256 u = m.to_canonical();
    memcpy(out + i * 32, u.v, 32);
  }
  return G16_OK;
}

}  // extern "C"
