//! src/callable_loci/gpu.rs -- the small amount of Rust around the FFI: the engine handle, the
//! structure-of-arrays tile, and the admission rule.  Not compiled in this repository.
use std::collections::VecDeque;
use std::error::Error;
use std::ffi::CStr;

use rust_htslib::bam::record::Record;

use super::ffi::*;
use super::options::CallableOptions;

/// One engine context per GPU, created once per run and reused for every contig.
pub struct GpuEngine { pub ctx: *mut ClCtx }

impl GpuEngine {
    pub fn new(o: &CallableOptions, device_id: i32) -> Result<Self, Box<dyn Error>> {
        let opt = ClOptions {
            min_depth: o.min_depth, max_depth: o.max_depth,
            min_mapping_quality: o.min_mapping_quality, min_base_quality: o.min_base_quality,
            min_depth_for_low_mapq: o.min_depth_for_low_mapq, max_low_mapq: o.max_low_mapq,
            max_low_mapq_fraction: o.max_low_mapq_fraction,
        };
        let mut ctx: *mut ClCtx = std::ptr::null_mut();
        let rc = unsafe { cl_create(&opt, device_id, std::ptr::null_mut(), &mut ctx) };
        if rc != 0 { return Err("no usable HIP device (the engine has no CPU fallback)".into()); }
        Ok(GpuEngine { ctx })
    }
}
impl Drop for GpuEngine {
    fn drop(&mut self) { unsafe { cl_destroy(self.ctx) } }
}

/// A negative cl_status becomes the error string the caller wraps as "Error processing contig: {e}".
pub fn check(gpu: &GpuEngine, rc: i32) -> Result<(), Box<dyn Error>> {
    if rc == 0 { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(cl_last_error(gpu.ctx)) }.to_string_lossy().into_owned();
    Err(msg.into())
}

/// The fields of a record the path consumes, appended read by read (cl_read_tile of callable_loci.h).
#[derive(Default)]
pub struct ReadSoa {
    pos: Vec<i32>, mapq: Vec<u8>,
    cigar_off: Vec<u32>, cigar: Vec<u32>,
    qual_off: Vec<u64>, qual: Vec<u8>,
}
impl ReadSoa {
    pub fn push(&mut self, pos: i32, mapq: u8, raw_cigar: &[u32], qual: &[u8]) {
        if self.cigar_off.is_empty() { self.cigar_off.push(0); self.qual_off.push(0); }
        self.pos.push(pos); self.mapq.push(mapq);
        self.cigar.extend_from_slice(raw_cigar);         // BAM encoding len<<4|op
        self.qual.extend_from_slice(qual);               // raw Phred, 0xFF when absent
        self.cigar_off.push(self.cigar.len() as u32);
        self.qual_off.push(self.qual.len() as u64);
    }
    pub fn tile(&self) -> ClReadTile {
        ClReadTile {
            n_reads: self.pos.len() as u64,
            pos: self.pos.as_ptr(), mapq: self.mapq.as_ptr(),
            cigar_off: self.cigar_off.as_ptr(), cigar: self.cigar.as_ptr(),
            qual_off: self.qual_off.as_ptr(), qual: self.qual.as_ptr(),
        }
    }
}

/// What htslib's bam_plp_push keeps (SURVEY 8a-11 (2),(7)): BAM_FUNMAP reads are dropped; a read that is not
/// the first at its start position is dropped while `maxcnt` reads are still listed, i.e. reads pushed
/// earlier whose end is not before that start position (they are freed lazily, one column late).
pub struct Admission {
    maxcnt: usize,
    tid: usize,
    live_ends: VecDeque<i64>,       // ends of the listed reads, ascending
    cur_start: i64,
    any: bool,
}
impl Admission {
    pub fn new(max_depth: u32, tid: usize) -> Self {
        // mod.rs:56-60: set_max_depth(max_depth) when it is > 0, else 500
        let maxcnt = if max_depth > 0 { max_depth as usize } else { 500 };
        Admission { maxcnt, tid, live_ends: VecDeque::new(), cur_start: -1, any: false }
    }
    /// true = the pileup holds this read and it spans reference positions
    pub fn accept(&mut self, rec: &Record) -> bool {
        if rec.is_unmapped() { return false; }
        let p = rec.pos();
        let end = rec.cigar().end_pos();                 // pos + reference length of the CIGAR
        let appended;
        if !self.any {
            appended = end > 0 || self.tid > 0;          // the iterator starts at (tid 0, pos 0)
            self.any = true; self.cur_start = p;
        } else if p == self.cur_start {
            if self.live_ends.len() >= self.maxcnt { return false; }
            appended = end > p;
        } else {
            self.cur_start = p;
            while self.live_ends.front().map_or(false, |&e| e < p) { self.live_ends.pop_front(); }
            appended = true;
        }
        if !appended { return false; }
        let at = self.live_ends.iter().rposition(|&e| e <= end).map_or(0, |i| i + 1);
        self.live_ends.insert(at, end);
        end > p
    }
}
