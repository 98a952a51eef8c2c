//! `extern "C"` declarations of include/g16_amd.h + include/g16_loaders.h, one for one.
//! tests/test_rust_shim.py (repository root) parses this file and asserts that the set of
//! functions and their arities equal the C headers'.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_float, c_int, c_void};

pub type g16_status = c_int;
pub const G16_OK: g16_status = 0;
pub const G16_ERR_INVALID: g16_status = 1;
pub const G16_ERR_DOMAIN_TOO_LARGE: g16_status = 2;
pub const G16_ERR_HIP: g16_status = 3;
pub const G16_ERR_NO_DEVICE: g16_status = 4;
pub const G16_ERR_IO: g16_status = 5;
pub const G16_ERR_INTERNAL: g16_status = 6;

pub const G16_PROOF_BYTES: usize = 256;
pub const G16_PARTIAL_BYTES: usize = 1024;
pub const G16_N_STAGES: usize = 10;
pub const G16_SHARD_AUTO: c_int = 0;
pub const G16_SHARD_POINTS: c_int = 1;
pub const G16_SHARD_BUCKETS: c_int = 2;
pub const G16_REDUCTION_CIRCOM: c_int = 0;
pub const G16_REDUCTION_LIBSNARK: c_int = 1;
pub const G16_QUERY_A: c_int = 0;
pub const G16_QUERY_B1: c_int = 1;
pub const G16_QUERY_L: c_int = 2;
pub const G16_QUERY_H: c_int = 3;

#[repr(C)]
pub struct g16_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct g16_setup {
    _private: [u8; 0],
}
#[repr(C)]
pub struct g16_zkey {
    _private: [u8; 0],
}
#[repr(C)]
pub struct g16_r1cs {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct g16_csr {
    pub row_ptr: *const u32,
    pub col: *const u32,
    pub coeff: *const u64,
    pub nnz: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct g16_key_desc {
    pub n_vars: u32,
    pub n_public: u32,
    pub domain_size: u32,
    pub a_query: *const u8,
    pub b_g1_query: *const u8,
    pub b_g2_query: *const u8,
    pub l_query: *const u8,
    pub h_query: *const u8,
    pub alpha_g1: [u8; 64],
    pub beta_g1: [u8; 64],
    pub delta_g1: [u8; 64],
    pub beta_g2: [u8; 128],
    pub delta_g2: [u8; 128],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct g16_options {
    pub device: c_int,
    pub rank: c_int,
    pub world: c_int,
    pub window_bits: c_int,
    pub planes: c_int,
    pub dist_wm: c_int,
    pub reduction: c_int,
    /// world > 1 / multi-device: G16_SHARD_AUTO, G16_SHARD_POINTS or G16_SHARD_BUCKETS
    pub shard: c_int,
    /// small keys: 0 = automatic fixed-base tables, > 0 require, < 0 never (include/g16_amd.h)
    pub fixed_tables: c_int,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct g16_vk_desc {
    pub alpha_g1: [u8; 64],
    pub beta_g2: [u8; 128],
    pub gamma_g2: [u8; 128],
    pub delta_g2: [u8; 128],
    pub ic: *const u8,
    pub ic_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct g16_zkey_header {
    pub n8q: u32,
    pub n8r: u32,
    pub q: [u8; 32],
    pub r: [u8; 32],
    pub n_vars: u32,
    pub n_public: u32,
    pub domain_size: u32,
    pub power: u32,
    pub alpha_g1: [u8; 64],
    pub beta_g1: [u8; 64],
    pub beta_g2: [u8; 128],
    pub gamma_g2: [u8; 128],
    pub delta_g1: [u8; 64],
    pub delta_g2: [u8; 128],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct g16_matrices {
    pub num_instance_variables: u32,
    pub num_witness_variables: u32,
    pub num_constraints: u32,
    pub a_num_non_zero: u64,
    pub b_num_non_zero: u64,
    pub a: g16_csr,
    pub b: g16_csr,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct g16_r1cs_header {
    pub version: u32,
    pub field_size: u32,
    pub prime: [u8; 32],
    pub n_wires: u32,
    pub n_pub_out: u32,
    pub n_pub_in: u32,
    pub n_prv_in: u32,
    pub n_labels: u64,
    pub n_constraints: u32,
    pub num_inputs: u32,
    pub num_aux: u32,
    pub num_variables: u32,
}

extern "C" {
    // ---- include/g16_amd.h ---------------------------------------------------------------------
    pub fn g16_ctx_create(key: *const g16_key_desc, a: *const g16_csr, b: *const g16_csr, num_constraints: u32, opt: *const g16_options, out: *mut *mut g16_ctx) -> g16_status;
    pub fn g16_ctx_create_sibling(
        donor: *mut g16_ctx,
        key: *const g16_key_desc,
        a: *const g16_csr,
        b: *const g16_csr,
        num_constraints: u32,
        opt: *const g16_options,
        out: *mut *mut g16_ctx,
    ) -> c_int;
    pub fn g16_ctx_destroy(ctx: *mut g16_ctx);
    pub fn g16_last_error(ctx: *const g16_ctx) -> *const c_char;
    pub fn g16_ctx_create_multi(key: *const g16_key_desc, a: *const g16_csr, b: *const g16_csr, num_constraints: u32, device_ids: *const c_int, n_dev: c_int, opt: *const g16_options, out: *mut *mut g16_ctx) -> g16_status;
    pub fn g16_witness_map(ctx: *mut g16_ctx, w: *const u64, n_vars: usize, h_out: *mut u64) -> g16_status;
    pub fn g16_msm_g1(ctx: *mut g16_ctx, which: c_int, scalars: *const u64, len: usize, out: *mut u8) -> g16_status;
    pub fn g16_msm_g2(ctx: *mut g16_ctx, scalars: *const u64, len: usize, out: *mut u8) -> g16_status;
    pub fn g16_witness_map_dev(ctx: *mut g16_ctx, w_dev: *const c_void, n_vars: usize, h_dev_out: *mut c_void) -> g16_status;
    pub fn g16_msm_g1_dev(ctx: *mut g16_ctx, which: c_int, scalars_dev: *const c_void, len: usize, out: *mut u8) -> g16_status;
    pub fn g16_msm_g2_dev(ctx: *mut g16_ctx, scalars_dev: *const c_void, len: usize, out: *mut u8) -> g16_status;
    pub fn g16_prove(ctx: *mut g16_ctx, r: *const u64, s: *const u64, w: *const u64, n_vars: usize, proof_out: *mut u8) -> g16_status;
    pub fn g16_prove_dev(ctx: *mut g16_ctx, r: *const u64, s: *const u64, w_dev: *const c_void, n_vars: usize, proof_out: *mut u8) -> g16_status;
    pub fn g16_prove_partial(ctx: *mut g16_ctx, r: *const u64, s: *const u64, w: *const u64, n_vars: usize, partial_out: *mut u8) -> g16_status;
    pub fn g16_prove_partial_dev(ctx: *mut g16_ctx, r: *const u64, s: *const u64, w_dev: *const c_void, n_vars: usize, partial_out: *mut u8) -> g16_status;
    pub fn g16_prove_finish(ctx: *mut g16_ctx, r: *const u64, s: *const u64, partials: *const u8, world: c_int, proof_out: *mut u8) -> g16_status;
    pub fn g16_dist_set_exchange_stream(ctx: *mut g16_ctx, hip_stream: *mut c_void, enabled: c_int) -> g16_status;
    pub fn g16_partial_buffer(ctx: *mut g16_ctx) -> *mut c_void;
    pub fn g16_gather_buffer(ctx: *mut g16_ctx) -> *mut c_void;
    pub fn g16_prove_finish_dev(ctx: *mut g16_ctx, r: *const u64, s: *const u64, proof_out: *mut u8) -> g16_status;
    pub fn g16_dist_exchange_bytes(ctx: *const g16_ctx) -> usize;
    pub fn g16_prove_dist_phase1(ctx: *mut g16_ctx, r: *const u64, s: *const u64, w_dev: *const c_void, n_vars: usize, send_dev: *mut c_void) -> g16_status;
    pub fn g16_prove_dist_phase2(ctx: *mut g16_ctx, recv_dev: *const c_void, send_dev: *mut c_void) -> g16_status;
    pub fn g16_prove_dist_phase3(ctx: *mut g16_ctx, recv_dev: *const c_void, partial_out: *mut u8) -> g16_status;
    pub fn g16_set_profiling(ctx: *mut g16_ctx, enabled: c_int) -> g16_status;
    pub fn g16_stage_times(ctx: *mut g16_ctx, ms: *mut c_float, launches: *mut u32) -> g16_status;
    pub fn g16_stage_name(stage: c_int) -> *const c_char;
    pub fn g16_ctx_info(ctx: *const g16_ctx, out: *mut u32) -> g16_status;
    pub fn g16_multi_links(ctx: *const g16_ctx, gbps: *mut f32, echo_us: *mut f32, cap: c_int, probe_bytes: *mut u64) -> g16_status;
    pub fn g16_witness_buffer(ctx: *mut g16_ctx) -> *mut c_void;
    pub fn g16_witness_upload(ctx: *mut g16_ctx, w: *const u64, n_vars: usize) -> g16_status;
    pub fn g16_witness_host_buffer(ctx: *mut g16_ctx) -> *mut c_void;
    pub fn g16_check_satisfied(device: c_int, a: *const g16_csr, b: *const g16_csr, c: *const g16_csr, num_constraints: u32, w: *const u64, n_vars: usize, first_unsatisfied: *mut i64) -> g16_status;
    pub fn g16_verify_batch(device: c_int, vk: *const g16_vk_desc, proofs: *const u8, public_inputs: *const u64, n_proofs: u32, ok_out: *mut u8) -> g16_status;
    pub fn g16_dist_attach_rccl(ctx: *mut g16_ctx, nccl_comm: *mut c_void) -> g16_status;
    pub fn g16_dist_rccl_ranks(ctx: *const g16_ctx) -> c_int;
    pub fn g16_prove_dist(ctx: *mut g16_ctx, r: *const u64, s: *const u64, w_dev: *const c_void, n_vars: usize, proof_out: *mut u8) -> g16_status;
    pub fn g16_fft_in_place(device: c_int, data: *mut u64, log_n: c_int, inverse: c_int, impl_: c_int) -> g16_status;
    pub fn g16_setup_create(device: c_int, at: *const g16_csr, bt: *const g16_csr, ct: *const g16_csr, n_vars: u32, n_public: u32, num_constraints: u32, toxic: *const u64, out: *mut *mut g16_setup) -> g16_status;
    pub fn g16_setup_create_ex(device: c_int, at: *const g16_csr, bt: *const g16_csr, ct: *const g16_csr, n_vars: u32, n_public: u32, num_constraints: u32, toxic: *const u64, reduction: c_int, out: *mut *mut g16_setup) -> g16_status;
    pub fn g16_setup_key(s: *mut g16_setup, key: *mut g16_key_desc, ic: *mut *const u8, ic_count: *mut u32, gamma_g2: *mut u8) -> g16_status;
    pub fn g16_setup_destroy(s: *mut g16_setup);

    // ---- include/g16_loaders.h -----------------------------------------------------------------
    pub fn g16_loader_last_error() -> *const c_char;
    pub fn g16_zkey_open(path: *const c_char, out: *mut *mut g16_zkey) -> g16_status;
    pub fn g16_zkey_open_mem(data: *const u8, len: usize, out: *mut *mut g16_zkey) -> g16_status;
    pub fn g16_zkey_close(z: *mut g16_zkey);
    pub fn g16_zkey_header_get(z: *const g16_zkey, out: *mut g16_zkey_header) -> g16_status;
    pub fn g16_zkey_key(z: *const g16_zkey, out: *mut g16_key_desc) -> g16_status;
    pub fn g16_zkey_ic(z: *const g16_zkey, count: *mut u32) -> *const u8;
    pub fn g16_zkey_matrices(z: *mut g16_zkey, out: *mut g16_matrices) -> g16_status;
    pub fn g16_zkey_write(path: *const c_char, key: *const g16_key_desc, ic: *const u8, gamma_g2: *const u8, a: *const g16_csr, b: *const g16_csr, num_constraints: u32) -> g16_status;
    pub fn g16_r1cs_open(path: *const c_char, out: *mut *mut g16_r1cs) -> g16_status;
    pub fn g16_r1cs_open_mem(data: *const u8, len: usize, out: *mut *mut g16_r1cs) -> g16_status;
    pub fn g16_r1cs_close(r: *mut g16_r1cs);
    pub fn g16_r1cs_header_get(r: *const g16_r1cs, out: *mut g16_r1cs_header) -> g16_status;
    pub fn g16_r1cs_matrices(r: *const g16_r1cs, a: *mut g16_csr, b: *mut g16_csr, c: *mut g16_csr) -> g16_status;
    pub fn g16_r1cs_wire_mapping(r: *const g16_r1cs, count: *mut u32) -> *const u64;
    pub fn g16_wtns_read(path: *const c_char, out: *mut *mut u64, n: *mut u32) -> g16_status;
    pub fn g16_wtns_read_mem(data: *const u8, len: usize, out: *mut *mut u64, n: *mut u32) -> g16_status;
    pub fn g16_free(p: *mut c_void);
    pub fn g16_fr_from_canonical(input: *const u8, out: *mut u64, n: usize) -> g16_status;
    pub fn g16_fr_to_canonical(input: *const u64, out: *mut u8, n: usize) -> g16_status;
}
