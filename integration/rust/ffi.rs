//! src/callable_loci/ffi.rs -- one declaration per symbol of include/callable_loci.h that the
//! coverage path uses.  Not compiled in this repository (no Rust toolchain in the image).
use std::os::raw::{c_char, c_int, c_uint, c_void};

#[repr(C)]
pub struct ClOptions {            // CallableOptions, options.rs:2-9
    pub min_depth: u32, pub max_depth: u32,
    pub min_mapping_quality: u8, pub min_base_quality: u8,
    pub min_depth_for_low_mapq: u32, pub max_low_mapq: u8,
    pub max_low_mapq_fraction: f64,
}
#[repr(C)]
pub struct ClReadTile {
    pub n_reads: u64,
    pub pos: *const i32, pub mapq: *const u8,
    pub cigar_off: *const u32, pub cigar: *const u32,
    pub qual_off: *const u64, pub qual: *const u8,
}
#[repr(C)]
pub struct ClReadTileBits {       // cl_read_tile_bits: the packed pass-bitmask variant
    pub n_reads: u64,
    pub pos: *const i32, pub mapq: *const u8,
    pub cigar_off: *const u32, pub cigar: *const u32,
    pub qual_off: *const u64,     // bit offsets into pass_bits
    pub pass_bits: *const u64,    // bit g = quality value g >= min_base_quality (mod.rs:33)
    pub pass_sum: *const u32,     // per read: sum of the passing values over its M/=/X bases (contig_profiler.rs:65-70)
}
#[repr(C)] #[derive(Default)]
pub struct ClContigSummary {
    pub state_counts: [u64; 6],   // indexed by CalledState as usize (types.rs:36-43)
    pub n_covered_bases: u64, pub summed_coverage: u64, pub summed_baseq: u64,
    pub summed_mapq: u64, pub quality_bases: u64,
    pub extent: u64, pub max_raw_depth: u64, pub n_intervals: u64,
}
#[repr(C)] pub struct ClInterval { pub start: u32, pub end: u32, pub state: u32 }
pub enum ClCtx {}

extern "C" {
    pub fn cl_create(opt: *const ClOptions, device_id: c_int, stream: *mut c_void, out: *mut *mut ClCtx) -> c_int;
    pub fn cl_destroy(ctx: *mut ClCtx);
    pub fn cl_last_error(ctx: *const ClCtx) -> *const c_char;
    pub fn cl_contig_begin(ctx: *mut ClCtx, tid: i32, contig_len: u32, ref_bases: *const u8, ref_len: u64) -> c_int;
    pub fn cl_contig_reserve(ctx: *mut ClCtx, n_reads: u64, n_cigar_ops: u64, n_qual_bytes: u64) -> c_int; // optional hint
    pub fn cl_contig_prefetch_qual(ctx: *mut ClCtx, qual: *const u8, n_bytes: u64) -> c_int;             // optional overlap
    pub fn cl_push_reads(ctx: *mut ClCtx, tile: *const ClReadTile) -> c_int;
    pub fn cl_push_reads_bits(ctx: *mut ClCtx, tile: *const ClReadTileBits) -> c_int;                     // optional: the test taken by the caller
    pub fn cl_contig_finish(ctx: *mut ClCtx, out: *mut ClContigSummary,
                            iv: *mut *const ClInterval, n_iv: *mut usize) -> c_int;
    pub fn cl_contig_abort(ctx: *mut ClCtx) -> c_int;      // error path: cancels an unclaimed prefetch, drops staged reads
    pub fn cl_contig_layout(ctx: *mut ClCtx, out: *mut ClLayoutInfo) -> c_int;   // what is resident: form, rows, HBM held
    // include/dut_bam.h: files in, files out, over one or several devices of a node (no collective: one process)
    pub fn dut_coverage_files_multi(bam: *const c_char, fasta: *const c_char, bed: *const c_char,
                                    summary_json: *const c_char, summary_html: *const c_char, opt: *const ClOptions,
                                    contigs: *const *const c_char, n_contigs: usize, devices: *const c_int, n_devices: usize,
                                    flags: c_uint, err: *mut c_char, err_len: usize) -> c_int;
}

#[repr(C)] #[derive(Default)]
pub struct ClLayoutInfo {         // cl_layout_info, include/callable_loci.h
    pub form: i32, pub counter_planes: u32,
    pub n_reads: u64, pub n_records: u64, pub n_windows: u64, pub n_qual: u64, pub n_cigar: u64,
    pub row_groups: u64, pub max_groups: u64, pub run_table_entries: u64,
    pub device_bytes: u64, pub upload_h2d_bytes: u64,
}
