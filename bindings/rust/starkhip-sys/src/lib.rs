//! FFI to `libstarkhip.so` (C ABI: `include/starkhip.h`) and a thin safe layer shaped like the calls it replaces in
//! Electron-Labs/starky_bls12_381:
//!
//! | reference (src/aggregate_proof.rs)                                            | here                         |
//! |-------------------------------------------------------------------------------|------------------------------|
//! | `StarkConfig::standard_fast_config()` + `rate_bits` override (:32-33,155-156) | [`Config::for_air`]          |
//! | `prove::<F, C, S, D>(stark, &config, trace, &pis, &mut timing)` (:59-65, ...) | [`Prover::prove_rows`], [`Prover::prove_recorded`] |
//! | `verify_stark_proof(stark, proof.clone(), &config)` (:67,113,146,177)         | [`verify`]                   |
//! | `S::generate_trace(..)` (+ public inputs built by the `*_main` drivers)       | [`record_final_exp`] etc. (or keep the reference's own generator and pass its rows) |
//!
//! NOT COMPILED in the image this repository is built in (no Rust toolchain there); kept in step with the header by hand.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_long, c_uint, c_void};

// ------------------------------------------------------------------------------------------------ raw declarations
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct starkhip_config_t {
    pub security_bits: u32,
    pub num_challenges: u32,
    pub rate_bits: u32,
    pub cap_height: u32,
    pub proof_of_work_bits: u32,
    pub arity_bits: u32,
    pub final_poly_bits: u32,
    pub num_query_rounds: u32,
}

/// `starkhip_air_t`: the `S: Stark` type parameter of the reference as a value.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Air {
    Fp12Mul = 0,        // FP12MulStark             src/fp12_mul.rs
    PairingPrecomp = 1, // PairingPrecompStark      src/calc_pairing_precomp.rs
    MillerLoop = 2,     // MillerLoopStark          src/miller_loop.rs
    FinalExp = 3,       // FinalExponentiateStark   src/final_exponentiate.rs
    EccAggregate = 4,   // ECCAggStark              src/ecc_aggregate.rs
}

pub const STARKHIP_OK: c_int = 0;
pub const STARKHIP_ERR_QUOTIENT_NOT_DIVISIBLE: c_int = -1;
pub const STARKHIP_ERR_ZETA_IN_SUBGROUP: c_int = -2;
pub const STARKHIP_ERR_BAD_SHAPE: c_int = -3;
pub const STARKHIP_ERR_HIP: c_int = -4;
pub const STARKHIP_ERR_OOM: c_int = -5;
pub const STARKHIP_ERR_NO_DEVICE: c_int = -6;
pub const STARKHIP_ERR_VERIFY: c_int = -7;
pub const STARKHIP_ERR_BAD_AIR: c_int = -8;
pub const STARKHIP_POW_SEARCH: u64 = u64::MAX;
pub const STARKHIP_N_PHASES: usize = 11;

/// `starkhip_proof_layout_t`: word offsets of every field of the proof blob.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct starkhip_proof_layout_t {
    pub n_columns: usize,
    pub n_quotient_polys: usize,
    pub degree_bits: usize,
    pub rate_bits: usize,
    pub cap_height: usize,
    pub n_fri_layers: usize,
    pub n_query_rounds: usize,
    pub final_poly_len: usize,
    pub n_public_inputs: usize,
    pub arity_bits: usize,
    pub off_trace_cap: usize,
    pub off_quotient_cap: usize,
    pub off_local_values: usize,
    pub off_next_values: usize,
    pub off_quotient_openings: usize,
    pub off_fri_caps: usize,
    pub off_query_rounds: usize,
    pub query_round_words: usize,
    pub off_final_poly: usize,
    pub off_pow_witness: usize,
    pub off_public_inputs: usize,
    pub total_words: usize,
    pub q_trace_leaf: usize,
    pub q_trace_siblings: usize,
    pub q_quotient_leaf: usize,
    pub q_quotient_siblings: usize,
    pub initial_sibling_count: usize,
    pub q_step_evals: [usize; 16],
    pub q_step_siblings: [usize; 16],
    pub step_sibling_count: [usize; 16],
}

/// `starkhip_pool_config_t` (0 = the library's default).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct starkhip_pool_config_t {
    pub device: c_int,
    pub big_contexts: c_uint,
    pub small_contexts: c_uint,
    pub generator_threads: c_uint,
    pub trace_threads: c_uint,
    pub commit_policy: c_uint,
    pub stream_priority: c_uint,
    pub warm_up: c_uint,
    pub gather_ms: f32,
}

/// `starkhip_ticket_info_t`: phase times of a pool proof and the job's timeline (seconds since the pool was created).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct starkhip_ticket_info_t {
    pub phase_ms: [f32; 11],
    pub kernel_ms: [f32; 3],
    pub host_ms: [f32; 2],
    pub t_submit: f64,
    pub t_generate_start: f64,
    pub t_generate_end: f64,
    pub t_prove_start: f64,
    pub t_done: f64,
    /// 0 quad form, 1 row form, 2 merged with other proofs' commitments, 3 lane form (`kernel_ms[1]` is that kernel's own duration)
    pub leaf_hash_form: c_int,
    pub leaf_hash_group: c_uint,
}

extern "C" {
    pub fn starkhip_config_standard_fast(cfg: *mut starkhip_config_t);
    pub fn starkhip_config_for_air(air: Air, cfg: *mut starkhip_config_t) -> c_int;
    pub fn starkhip_air_columns(air: Air) -> c_int;
    pub fn starkhip_air_public_inputs(air: Air) -> c_int;
    pub fn starkhip_air_constraint_degree(air: Air) -> c_int;
    pub fn starkhip_air_num_constraints(air: Air) -> c_int;
    pub fn starkhip_air_default_rows(air: Air) -> c_int;

    pub fn starkhip_trace_fp12_mul(x: *const u32, y: *const u32, trace: *mut u64, n_rows: usize, public_inputs: *mut u64) -> c_int;
    pub fn starkhip_trace_final_exp(x: *const u32, trace: *mut u64, n_rows: usize, public_inputs: *mut u64) -> c_int;
    pub fn starkhip_trace_miller_loop(px: *const u32, py: *const u32, qx: *const u32, qy: *const u32, qz: *const u32, trace: *mut u64,
                                      n_rows: usize, public_inputs: *mut u64) -> c_int;
    pub fn starkhip_trace_pairing_precomp(qx: *const u32, qy: *const u32, qz: *const u32, trace: *mut u64, n_rows: usize,
                                          public_inputs: *mut u64) -> c_int;
    pub fn starkhip_trace_ecc_aggregate(points: *const u32, bits: *const u8, trace: *mut u64, n_rows: usize, public_inputs: *mut u64) -> c_int;

    pub fn starkhip_init(device_ordinal: c_int, ctx: *mut *mut c_void) -> c_int;
    pub fn starkhip_shutdown(ctx: *mut c_void);
    pub fn starkhip_set_option(ctx: *mut c_void, name: *const c_char, value: c_long) -> c_int;
    pub fn starkhip_prove(ctx: *mut c_void, air: Air, cfg: *const starkhip_config_t, trace: *const u64, n_rows: usize, n_cols: usize,
                          trace_layout: c_int, trace_on_device: c_int, public_inputs: *const u64, n_pis: usize, pow_witness: u64,
                          proof: *mut *mut u64, proof_words: *mut usize) -> c_int;

    pub fn starkhip_trace_log_begin(log: *mut *mut c_void) -> c_int;
    pub fn starkhip_trace_set_threads(n: c_int) -> c_int;
    pub fn starkhip_trace_log_end(log: *mut c_void) -> c_int;
    pub fn starkhip_trace_log_free(log: *mut c_void);
    pub fn starkhip_trace_log_info(log: *const c_void, n_rows: *mut usize, n_cols: *mut usize, n_records: *mut usize, n_words: *mut usize) -> c_int;
    pub fn starkhip_prove_compact(ctx: *mut c_void, air: Air, cfg: *const starkhip_config_t, log: *const c_void, public_inputs: *const u64,
                                  n_pis: usize, pow_witness: u64, proof: *mut *mut u64, proof_words: *mut usize) -> c_int;

    pub fn starkhip_pool_create(cfg: *const starkhip_pool_config_t, pool: *mut *mut c_void) -> c_int;
    pub fn starkhip_pool_destroy(pool: *mut c_void);
    pub fn starkhip_pool_submit(pool: *mut c_void, air: Air, cfg: *const starkhip_config_t, trace: *const u64, n_rows: usize, n_cols: usize,
                                trace_layout: c_int, trace_on_device: c_int, public_inputs: *const u64, n_pis: usize, pow_witness: u64,
                                ticket: *mut u64) -> c_int;
    pub fn starkhip_prove_columns(ctx: *mut c_void, air: Air, cfg: *const starkhip_config_t, columns: *const *const u64, n_rows: usize, n_cols: usize,
                                  public_inputs: *const u64, n_pis: usize, pow_witness: u64, proof: *mut *mut u64, proof_words: *mut usize) -> c_int;
    pub fn starkhip_pool_submit_columns(pool: *mut c_void, air: Air, cfg: *const starkhip_config_t, columns: *const *const u64, n_rows: usize,
                                        n_cols: usize, public_inputs: *const u64, n_pis: usize, pow_witness: u64, ticket: *mut u64) -> c_int;
    pub fn starkhip_pool_submit_compact(pool: *mut c_void, air: Air, cfg: *const starkhip_config_t, log: *const c_void, public_inputs: *const u64,
                                        n_pis: usize, pow_witness: u64, ticket: *mut u64) -> c_int;
    pub fn starkhip_pool_submit_witness(pool: *mut c_void, air: Air, cfg: *const starkhip_config_t, operands: *const u32, n_limbs: usize,
                                        pow_witness: u64, ticket: *mut u64) -> c_int;
    pub fn starkhip_pool_wait(pool: *mut c_void, ticket: u64, proof: *mut *mut u64, proof_words: *mut usize, info: *mut starkhip_ticket_info_t) -> c_int;

    // one process, several devices: a pool per device behind one handle (`slot` = -1: the library places the job)
    pub fn starkhip_multipool_create(devices: *const c_int, n_devices: usize, cfg: *const starkhip_pool_config_t, mpool: *mut *mut c_void) -> c_int;
    pub fn starkhip_multipool_destroy(mpool: *mut c_void);
    pub fn starkhip_multipool_size(mpool: *const c_void) -> usize;
    pub fn starkhip_multipool_submit(mpool: *mut c_void, slot: c_int, air: Air, cfg: *const starkhip_config_t, trace: *const u64, n_rows: usize,
                                     n_cols: usize, trace_layout: c_int, trace_on_device: c_int, public_inputs: *const u64, n_pis: usize,
                                     pow_witness: u64, ticket: *mut u64) -> c_int;
    pub fn starkhip_multipool_submit_columns(mpool: *mut c_void, slot: c_int, air: Air, cfg: *const starkhip_config_t, columns: *const *const u64,
                                             n_rows: usize, n_cols: usize, public_inputs: *const u64, n_pis: usize, pow_witness: u64,
                                             ticket: *mut u64) -> c_int;
    pub fn starkhip_multipool_submit_witness(mpool: *mut c_void, slot: c_int, air: Air, cfg: *const starkhip_config_t, operands: *const u32,
                                             n_limbs: usize, pow_witness: u64, ticket: *mut u64) -> c_int;
    pub fn starkhip_multipool_submit_witness_batch(mpool: *mut c_void, n_jobs: usize, airs: *const Air, operands: *const *const u32,
                                                   n_limbs: *const usize, pow_witness: u64, tickets: *mut u64, rcs: *mut c_int) -> c_int;
    pub fn starkhip_multipool_ticket_slot(mpool: *const c_void, ticket: u64) -> c_int;
    pub fn starkhip_multipool_wait(mpool: *mut c_void, ticket: u64, proof: *mut *mut u64, proof_words: *mut usize,
                                   info: *mut starkhip_ticket_info_t) -> c_int;
    pub fn starkhip_hw_queues_status() -> c_int;

    pub fn starkhip_last_timings(ctx: *mut c_void, ms: *mut f32) -> c_int;
    pub fn starkhip_host_alloc(ctx: *mut c_void, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn starkhip_host_free(p: *mut c_void);

    pub fn starkhip_verify(air: Air, cfg: *const starkhip_config_t, proof: *const u64, proof_words: usize) -> c_int;
    pub fn starkhip_proof_layout(proof: *const u64, proof_words: usize, out: *mut starkhip_proof_layout_t) -> c_int;
    pub fn starkhip_free(p: *mut c_void);
    pub fn starkhip_proof_blob_stats(out: *mut u64);
    pub fn starkhip_error_string(code: c_int) -> *const c_char;
}

// ------------------------------------------------------------------------------------------------ safe layer
/// The reference's errors: `prove` returns `anyhow::Result`; an invalid witness is "Quotient has failed, the vanishing
/// polynomial is not divisible by Z_H", a bad challenge "Opening point is in the subgroup".
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct Error(pub c_int);

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        let s = unsafe { std::ffi::CStr::from_ptr(starkhip_error_string(self.0)) };
        write!(f, "starkhip: {} ({})", s.to_string_lossy(), self.0)
    }
}
impl std::error::Error for Error {}

fn check(rc: c_int) -> Result<(), Error> {
    if rc == STARKHIP_OK { Ok(()) } else { Err(Error(rc)) }
}

pub type Config = starkhip_config_t;
impl Config {
    /// `StarkConfig::standard_fast_config()` with the `rate_bits` the reference's driver sets for this AIR.
    pub fn for_air(air: Air) -> Result<Config, Error> {
        let mut cfg = Config::default();
        check(unsafe { starkhip_config_for_air(air, &mut cfg) })?;
        Ok(cfg)
    }
}

/// The proof blob (field order of `StarkProofWithPublicInputs`); `layout()` gives the offset of every field.
pub struct Proof(pub Vec<u64>);
impl Proof {
    pub fn layout(&self) -> Result<starkhip_proof_layout_t, Error> {
        let mut l = std::mem::MaybeUninit::<starkhip_proof_layout_t>::zeroed();
        check(unsafe { starkhip_proof_layout(self.0.as_ptr(), self.0.len(), l.as_mut_ptr()) })?;
        Ok(unsafe { l.assume_init() })
    }
    pub fn public_inputs(&self) -> Result<&[u64], Error> {
        let l = self.layout()?;
        Ok(&self.0[l.off_public_inputs..l.off_public_inputs + l.n_public_inputs])
    }
}

/// A trace recorded as runs by one of the `record_*` functions (153 MB instead of 4.8 GB of rows for FinalExp).
pub struct RecordedTrace {
    log: *mut c_void,
    pub public_inputs: Vec<u64>,
}
unsafe impl Send for RecordedTrace {} // immutable after recording; may be proven from any thread
impl Drop for RecordedTrace {
    fn drop(&mut self) {
        unsafe { starkhip_trace_log_free(self.log) }
    }
}

fn record(air: Air, fill: impl FnOnce(*mut u64) -> c_int) -> Result<RecordedTrace, Error> {
    let n_pis = unsafe { starkhip_air_public_inputs(air) } as usize;
    let mut pis = vec![0u64; n_pis];
    let mut log = std::ptr::null_mut();
    check(unsafe { starkhip_trace_log_begin(&mut log) })?;
    let rc = fill(pis.as_mut_ptr());
    let end = unsafe { starkhip_trace_log_end(log) };
    if rc != STARKHIP_OK || end != STARKHIP_OK {
        unsafe { starkhip_trace_log_free(log) };
        return Err(Error(if rc != STARKHIP_OK { rc } else { end }));
    }
    Ok(RecordedTrace { log, public_inputs: pis })
}

/// `FinalExponentiateStark::generate_trace(x)` + the public inputs of `final_exponentiate_main` (src/aggregate_proof.rs:150-165).
pub fn record_final_exp(x: &[u32; 144]) -> Result<RecordedTrace, Error> {
    record(Air::FinalExp, |pis| unsafe { starkhip_trace_final_exp(x.as_ptr(), std::ptr::null_mut(), 8192, pis) })
}
/// `MillerLoopStark::generate_trace(..)` (src/aggregate_proof.rs:78-101).
pub fn record_miller_loop(px: &[u32; 12], py: &[u32; 12], qx: &[u32; 24], qy: &[u32; 24], qz: &[u32; 24]) -> Result<RecordedTrace, Error> {
    record(Air::MillerLoop, |pis| unsafe {
        starkhip_trace_miller_loop(px.as_ptr(), py.as_ptr(), qx.as_ptr(), qy.as_ptr(), qz.as_ptr(), std::ptr::null_mut(), 1024, pis)
    })
}
/// `PairingPrecompStark::generate_trace(..)` (src/aggregate_proof.rs:23-54).
pub fn record_pairing_precomp(qx: &[u32; 24], qy: &[u32; 24], qz: &[u32; 24]) -> Result<RecordedTrace, Error> {
    record(Air::PairingPrecomp, |pis| unsafe {
        starkhip_trace_pairing_precomp(qx.as_ptr(), qy.as_ptr(), qz.as_ptr(), std::ptr::null_mut(), 1024, pis)
    })
}
/// `FP12MulStark::generate_trace(x, y)` (src/aggregate_proof.rs:120-133).
pub fn record_fp12_mul(x: &[u32; 144], y: &[u32; 144]) -> Result<RecordedTrace, Error> {
    record(Air::Fp12Mul, |pis| unsafe { starkhip_trace_fp12_mul(x.as_ptr(), y.as_ptr(), std::ptr::null_mut(), 16, pis) })
}

/// One prover context = one HIP stream and its buffers on one GPU; one `prove` at a time per context, any number of contexts
/// per GPU (four in flight per GPU is the measured optimum for FinalExp).
pub struct Prover {
    ctx: *mut c_void,
}
unsafe impl Send for Prover {}
impl Drop for Prover {
    fn drop(&mut self) {
        unsafe { starkhip_shutdown(self.ctx) }
    }
}

impl Prover {
    pub fn new(device_ordinal: i32) -> Result<Prover, Error> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { starkhip_init(device_ordinal as c_int, &mut ctx) })?;
        Ok(Prover { ctx })
    }

    unsafe fn take(p: *mut u64, words: usize) -> Proof {
        let v = std::slice::from_raw_parts(p, words).to_vec();
        starkhip_free(p as *mut c_void);
        Proof(v)
    }

    /// `prove(stark, &config, trace_rows_to_poly_values(trace), &public_inputs, &mut timing)` for a trace as
    /// `generate_trace` returns it: rows of canonical Goldilocks cells (`GoldilocksField` is `repr(transparent)` over u64:
    /// pass `to_canonical_u64()` of every cell).
    pub fn prove_rows<const COLUMNS: usize>(&mut self, air: Air, cfg: &Config, trace: &[[u64; COLUMNS]], public_inputs: &[u64]) -> Result<Proof, Error> {
        let (mut p, mut w) = (std::ptr::null_mut::<u64>(), 0usize);
        check(unsafe {
            starkhip_prove(self.ctx, air, cfg, trace.as_ptr() as *const u64, trace.len(), COLUMNS, /* row-major */ 0, /* host */ 0,
                           public_inputs.as_ptr(), public_inputs.len(), STARKHIP_POW_SEARCH, &mut p, &mut w)
        })?;
        Ok(unsafe { Prover::take(p, w) })
    }

    /// `prove(stark, &config, trace_poly_values, &public_inputs, &mut timing)` with the reference's literal argument
    /// (src/aggregate_proof.rs:168-175): `trace_poly_values: Vec<PolynomialValues<F>>` is one heap allocation per column.  Pass
    /// each `PolynomialValues::values` as a `&[u64]` (canonical cells); the library gathers the scattered columns itself.
    pub fn prove_columns(&mut self, air: Air, cfg: &Config, columns: &[&[u64]], public_inputs: &[u64]) -> Result<Proof, Error> {
        let n_rows = columns.first().map_or(0, |c| c.len());
        if columns.iter().any(|c| c.len() != n_rows) {
            return Err(Error(STARKHIP_ERR_BAD_SHAPE));
        }
        let ptrs: Vec<*const u64> = columns.iter().map(|c| c.as_ptr()).collect();
        let (mut p, mut w) = (std::ptr::null_mut::<u64>(), 0usize);
        check(unsafe {
            starkhip_prove_columns(self.ctx, air, cfg, ptrs.as_ptr(), n_rows, ptrs.len(), public_inputs.as_ptr(), public_inputs.len(),
                                   STARKHIP_POW_SEARCH, &mut p, &mut w)
        })?;
        Ok(unsafe { Prover::take(p, w) })
    }

    /// The same proof, byte for byte, from a recorded trace (expanded on the device).
    pub fn prove_recorded(&mut self, air: Air, cfg: &Config, trace: &RecordedTrace) -> Result<Proof, Error> {
        let (mut p, mut w) = (std::ptr::null_mut::<u64>(), 0usize);
        check(unsafe {
            starkhip_prove_compact(self.ctx, air, cfg, trace.log, trace.public_inputs.as_ptr(), trace.public_inputs.len(), STARKHIP_POW_SEARCH,
                                   &mut p, &mut w)
        })?;
        Ok(unsafe { Prover::take(p, w) })
    }

    /// Milliseconds of the last proof's phases (upload, ifft_lde, trace_merkle, quotient, quotient_commit, openings, fri_combine,
    /// fri_commit, pow, queries, total) -- what the reference's `TimingTree` would print.
    pub fn last_timings(&mut self) -> Result<[f32; STARKHIP_N_PHASES], Error> {
        let mut ms = [0f32; STARKHIP_N_PHASES];
        check(unsafe { starkhip_last_timings(self.ctx, ms.as_mut_ptr()) })?;
        Ok(ms)
    }
}

/// The library's proof pool (`starkhip_pool_*`): what replaces the six back-to-back `prove` calls of
/// `generate_aggregate_proof` (src/aggregate_proof.rs:304-370).  `submit_*` returns at once with a ticket; the proofs of one
/// signature -- or of many -- are in flight together, traces are generated on the pool's threads, and the library decides
/// which Merkle commitments share a launch.
///
/// `Pool::new_multi(&[0, 1, .., 7], &cfg)` is the same object over SEVERAL devices (`starkhip_multipool_*`): the reference's
/// caller is one process (src/aggregate_proof.rs:304-370, :402-414), and this gives that process a pool per GPU behind the same
/// `submit_*` / `wait` -- every job goes to the pool with the least outstanding work, FinalExp jobs to the pool with the fewest
/// of them (longest processing time first); no process group, no collective.
pub struct Pool {
    pool: *mut c_void,
    multi: bool,
}
unsafe impl Send for Pool {}
unsafe impl Sync for Pool {} // submit / wait are thread-safe
impl Drop for Pool {
    fn drop(&mut self) {
        unsafe {
            if self.multi {
                starkhip_multipool_destroy(self.pool)
            } else {
                starkhip_pool_destroy(self.pool)
            }
        }
    }
}
/// A proof in flight.  The lifetime ties the ticket to the pool AND to everything the pool still reads for it: `submit_rows`
/// hands the library borrowed rows, which a prover thread uploads some time after `submit_rows` has returned (starkhip.h:
/// "must stay valid until the ticket has been waited for").  The borrow therefore lasts until `wait` consumes the ticket, and a
/// ticket that is dropped un-waited waits in its `Drop` (the proof is discarded), so safe code cannot free or mutate the rows
/// under the copy.  (`std::mem::forget` of a ticket leaks the job, and with it the library's right to read the rows: do not.)
pub struct Ticket<'a> {
    pub air: Air,
    pub id: u64,
    pool: &'a Pool,
    waited: bool,
    _inputs: std::marker::PhantomData<&'a [u64]>,
}
impl<'a> Drop for Ticket<'a> {
    fn drop(&mut self) {
        if !self.waited {
            // the library may still be reading the borrowed inputs: block until it is done with them, drop the proof
            unsafe { self.pool.raw_wait(self.id, std::ptr::null_mut(), std::ptr::null_mut()) };
        }
    }
}

impl Pool {
    pub fn new(cfg: &starkhip_pool_config_t) -> Result<Pool, Error> {
        let mut pool = std::ptr::null_mut();
        check(unsafe { starkhip_pool_create(cfg, &mut pool) })?;
        Ok(Pool { pool, multi: false })
    }
    /// One pool per entry of `devices` (an ordinal may repeat: two pools on one card), each configured by `cfg`
    /// (`cfg.device` is ignored), the process's CPU budget split between them.
    pub fn new_multi(devices: &[c_int], cfg: &starkhip_pool_config_t) -> Result<Pool, Error> {
        let mut pool = std::ptr::null_mut();
        check(unsafe { starkhip_multipool_create(devices.as_ptr(), devices.len(), cfg, &mut pool) })?;
        Ok(Pool { pool, multi: true })
    }
    /// Pools behind this handle (1 for `new`).
    pub fn devices(&self) -> usize {
        if self.multi { unsafe { starkhip_multipool_size(self.pool) } } else { 1 }
    }
    /// Which of them a ticket's job went to.
    pub fn slot_of(&self, ticket: &Ticket<'_>) -> usize {
        if self.multi { unsafe { starkhip_multipool_ticket_slot(self.pool, ticket.id) }.max(0) as usize } else { 0 }
    }
    unsafe fn raw_wait(&self, id: u64, p: *mut *mut u64, w: *mut usize) -> c_int {
        if self.multi {
            starkhip_multipool_wait(self.pool, id, p, w, std::ptr::null_mut())
        } else {
            starkhip_pool_wait(self.pool, id, p, w, std::ptr::null_mut())
        }
    }
    /// A whole batch of the reference's drivers at once -- BASELINE configs[3] / [4]: the six proofs of one signature, or the 48 of
    /// eight -- placed longest first over the devices (on one pool: in the given order).  Tickets come back in the jobs' order.
    pub fn submit_batch<'a>(&'a self, jobs: &[(Air, &[u32])]) -> Result<Vec<Ticket<'a>>, Error> {
        if !self.multi {
            return jobs.iter().map(|(air, ops)| self.submit(*air, ops)).collect();
        }
        let airs: Vec<Air> = jobs.iter().map(|j| j.0).collect();
        let ptrs: Vec<*const u32> = jobs.iter().map(|j| j.1.as_ptr()).collect();
        let lens: Vec<usize> = jobs.iter().map(|j| j.1.len()).collect();
        let mut ids = vec![0u64; jobs.len()];
        let rc = unsafe {
            starkhip_multipool_submit_witness_batch(self.pool, jobs.len(), airs.as_ptr(), ptrs.as_ptr(), lens.as_ptr(), STARKHIP_POW_SEARCH,
                                                    ids.as_mut_ptr(), std::ptr::null_mut())
        };
        // the jobs that were accepted are in flight: their tickets are made first, so that an error drops (= waits for) them
        let tickets: Vec<Ticket<'a>> = jobs.iter().zip(ids.iter()).filter(|(_, id)| **id != 0).map(|(j, id)| self.ticket(j.0, *id)).collect();
        check(rc)?;
        Ok(tickets)
    }
    fn ticket<'a>(&'a self, air: Air, id: u64) -> Ticket<'a> {
        Ticket { air, id, pool: self, waited: false, _inputs: std::marker::PhantomData }
    }
    /// generate_trace + prove of one of the reference's drivers (src/aggregate_proof.rs:23-179) from its operands as u32 limbs,
    /// packed as `starkhip.h` says (e.g. MillerLoop: px, py, qx, qy, qz).  The operands are copied: nothing stays borrowed but the pool.
    pub fn submit<'a>(&'a self, air: Air, operands: &[u32]) -> Result<Ticket<'a>, Error> {
        let mut t = 0u64;
        check(unsafe {
            if self.multi {
                starkhip_multipool_submit_witness(self.pool, -1, air, std::ptr::null(), operands.as_ptr(), operands.len(), STARKHIP_POW_SEARCH, &mut t)
            } else {
                starkhip_pool_submit_witness(self.pool, air, std::ptr::null(), operands.as_ptr(), operands.len(), STARKHIP_POW_SEARCH, &mut t)
            }
        })?;
        Ok(self.ticket(air, t))
    }
    /// The reference's own generator stays: hand over the rows `generate_trace` returned.  `trace` and `public_inputs` stay
    /// borrowed for as long as the ticket lives, i.e. until `wait` has returned (or the ticket has been dropped, which waits).
    pub fn submit_rows<'a, const COLUMNS: usize>(&'a self, air: Air, cfg: &Config, trace: &'a [[u64; COLUMNS]], public_inputs: &'a [u64]) -> Result<Ticket<'a>, Error> {
        let mut t = 0u64;
        check(unsafe {
            if self.multi {
                starkhip_multipool_submit(self.pool, -1, air, cfg, trace.as_ptr() as *const u64, trace.len(), COLUMNS, 0, 0, public_inputs.as_ptr(),
                                          public_inputs.len(), STARKHIP_POW_SEARCH, &mut t)
            } else {
                starkhip_pool_submit(self.pool, air, cfg, trace.as_ptr() as *const u64, trace.len(), COLUMNS, 0, 0, public_inputs.as_ptr(),
                                     public_inputs.len(), STARKHIP_POW_SEARCH, &mut t)
            }
        })?;
        Ok(self.ticket(air, t))
    }
    /// `prove(stark, &config, trace_poly_values, ..)` with the literal argument of src/aggregate_proof.rs:168-175: one heap
    /// allocation per column (`Vec<PolynomialValues<F>>`, F = GoldilocksField is `repr(transparent)` over u64).  The library
    /// gathers the scattered columns through its page-locked staging; the column vectors stay borrowed until `wait`.
    pub fn submit_columns<'a>(&'a self, air: Air, cfg: &Config, columns: &'a [Vec<u64>], public_inputs: &'a [u64]) -> Result<Ticket<'a>, Error> {
        let n_rows = columns.first().map_or(0, |c| c.len());
        if columns.iter().any(|c| c.len() != n_rows) {
            return Err(Error(STARKHIP_ERR_BAD_SHAPE));
        }
        // the pointer table is copied by the library before submit_columns returns
        let ptrs: Vec<*const u64> = columns.iter().map(|c| c.as_ptr()).collect();
        let mut t = 0u64;
        check(unsafe {
            if self.multi {
                starkhip_multipool_submit_columns(self.pool, -1, air, cfg, ptrs.as_ptr(), n_rows, ptrs.len(), public_inputs.as_ptr(),
                                                  public_inputs.len(), STARKHIP_POW_SEARCH, &mut t)
            } else {
                starkhip_pool_submit_columns(self.pool, air, cfg, ptrs.as_ptr(), n_rows, ptrs.len(), public_inputs.as_ptr(), public_inputs.len(),
                                             STARKHIP_POW_SEARCH, &mut t)
            }
        })?;
        Ok(self.ticket(air, t))
    }
    /// Blocks until the proof is done; the `Err` is what `prove` would have returned.
    pub fn wait(&self, mut ticket: Ticket<'_>) -> Result<Proof, Error> {
        let (mut p, mut w) = (std::ptr::null_mut::<u64>(), 0usize);
        ticket.waited = true; // starkhip_pool_wait consumes the ticket whatever it returns
        check(unsafe { self.raw_wait(ticket.id, &mut p, &mut w) })?;
        Ok(unsafe { Prover::take(p, w) })
    }
}

/// `verify_stark_proof(stark, proof, &config)` (CPU).
pub fn verify(air: Air, cfg: &Config, proof: &Proof) -> Result<(), Error> {
    check(unsafe { starkhip_verify(air, cfg, proof.0.as_ptr(), proof.0.len()) })
}

/// Example: the body of `final_exponentiate_main` (src/aggregate_proof.rs:150-179) on the GPU.
pub fn final_exponentiate_main(prover: &mut Prover, x: &[u32; 144]) -> Result<Proof, Error> {
    let cfg = Config::for_air(Air::FinalExp)?;
    let trace = record_final_exp(x)?;
    let proof = prover.prove_recorded(Air::FinalExp, &cfg, &trace)?;
    verify(Air::FinalExp, &cfg, &proof)?;
    Ok(proof)
}

#[allow(dead_code)]
fn _abi_sizes() {
    // the header's structs are plain C: eight u32 / 27 + 48 size_t
    let _ = [(); 32][std::mem::size_of::<starkhip_config_t>() - 32];
    let _: c_uint = 0;
}

// ------------------------------------------------------------------------------------------------ hand-off to the recursion
/// `StarkProofWithPublicInputs<F, C, D>` filled FIELD BY FIELD from the blob, for
/// `starky::recursive_verifier::set_stark_proof_with_pis_target` in `recursive_proof` (src/aggregate_proof.rs:435-439) -- no
/// serde names involved.  Compiled only with `--features plonky2` next to the reference's own plonky2 / starky git dependencies
/// (Cargo.toml of the reference, :9-10); this repository's image has neither the crates nor a Rust toolchain, so the struct and
/// field names below are restated from memory of starky 0.1.x @ 666f3151 and are UNPINNED (COMPAT.md section 6.2): a maintainer
/// compiles this module once against the real crates and fixes whatever name differs -- the offsets come from
/// `starkhip_proof_layout` and do not depend on names.
#[cfg(feature = "plonky2")]
pub mod handoff {
    use super::{Error, Proof};
    use plonky2::field::extension::quadratic::QuadraticExtension;
    use plonky2::field::goldilocks_field::GoldilocksField;
    use plonky2::field::polynomial::PolynomialCoeffs;
    use plonky2::field::types::Field;
    use plonky2::fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep};
    use plonky2::hash::hash_types::HashOut;
    use plonky2::hash::merkle_proofs::MerkleProof;
    use plonky2::hash::merkle_tree::MerkleCap;
    use plonky2::hash::poseidon::PoseidonHash;
    use plonky2::plonk::config::PoseidonGoldilocksConfig;
    use starky::proof::{StarkOpeningSet, StarkProof, StarkProofWithPublicInputs};

    type F = GoldilocksField;
    type FE = QuadraticExtension<F>;
    type C = PoseidonGoldilocksConfig;
    const D: usize = 2;

    fn f(w: u64) -> F { F::from_canonical_u64(w) }
    fn ext(w: &[u64]) -> FE { FE::from_basefield_array([f(w[0]), f(w[1])]) }
    fn exts(w: &[u64]) -> Vec<FE> { w.chunks_exact(2).map(ext).collect() }
    fn hash(w: &[u64]) -> HashOut<F> { HashOut { elements: [f(w[0]), f(w[1]), f(w[2]), f(w[3])] } }
    fn cap(w: &[u64]) -> MerkleCap<F, PoseidonHash> { MerkleCap(w.chunks_exact(4).map(hash).collect()) }
    fn path(w: &[u64]) -> MerkleProof<F, PoseidonHash> { MerkleProof { siblings: w.chunks_exact(4).map(hash).collect() } }

    pub fn stark_proof_with_public_inputs(proof: &Proof) -> Result<StarkProofWithPublicInputs<F, C, D>, Error> {
        let l = proof.layout()?;
        let b = &proof.0[..];
        let ncap = 4usize << l.cap_height; // words per cap
        let openings = StarkOpeningSet {
            local_values: exts(&b[l.off_local_values..l.off_local_values + 2 * l.n_columns]),
            next_values: exts(&b[l.off_next_values..l.off_next_values + 2 * l.n_columns]),
            permutation_zs: None,       // none of the five AIRs uses permutation arguments
            permutation_zs_next: None,
            quotient_polys: exts(&b[l.off_quotient_openings..l.off_quotient_openings + 2 * l.n_quotient_polys]),
        };
        let commit_phase_merkle_caps = (0..l.n_fri_layers).map(|i| cap(&b[l.off_fri_caps + i * ncap..l.off_fri_caps + (i + 1) * ncap])).collect();
        let query_round_proofs = (0..l.n_query_rounds)
            .map(|r| {
                let q = &b[l.off_query_rounds + r * l.query_round_words..l.off_query_rounds + (r + 1) * l.query_round_words];
                let d0 = 4 * l.initial_sibling_count;
                // oracle 0 = trace, oracle 1 = quotient polynomials (the order PolynomialBatch commitments are passed to prove_openings)
                let evals_proofs = vec![
                    (q[l.q_trace_leaf..l.q_trace_leaf + l.n_columns].iter().map(|&w| f(w)).collect(), path(&q[l.q_trace_siblings..l.q_trace_siblings + d0])),
                    (q[l.q_quotient_leaf..l.q_quotient_leaf + l.n_quotient_polys].iter().map(|&w| f(w)).collect(),
                     path(&q[l.q_quotient_siblings..l.q_quotient_siblings + d0])),
                ];
                let steps = (0..l.n_fri_layers)
                    .map(|s| FriQueryStep {
                        evals: exts(&q[l.q_step_evals[s]..l.q_step_evals[s] + (2usize << l.arity_bits)]),
                        merkle_proof: path(&q[l.q_step_siblings[s]..l.q_step_siblings[s] + 4 * l.step_sibling_count[s]]),
                    })
                    .collect();
                FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs }, steps }
            })
            .collect();
        let opening_proof = FriProof {
            commit_phase_merkle_caps,
            query_round_proofs,
            final_poly: PolynomialCoeffs::new(exts(&b[l.off_final_poly..l.off_final_poly + 2 * l.final_poly_len])),
            pow_witness: f(b[l.off_pow_witness]),
        };
        Ok(StarkProofWithPublicInputs {
            proof: StarkProof {
                trace_cap: cap(&b[l.off_trace_cap..l.off_trace_cap + ncap]),
                permutation_zs_cap: None,
                quotient_polys_cap: cap(&b[l.off_quotient_cap..l.off_quotient_cap + ncap]),
                openings,
                opening_proof,
            },
            public_inputs: b[l.off_public_inputs..l.off_public_inputs + l.n_public_inputs].iter().map(|&w| f(w)).collect(),
        })
    }
}
