//! The replacement of callable_loci::process_single_contig (src/callable_loci/mod.rs:44-147): same
//! signature plus the engine handle; the htslib pileup and the per-column calls are gone, everything else
//! (fetch, CallableProfiler, ContigProfiler fields, finish_contig) is as before.  Not compiled in this
//! repository.
use std::collections::{HashMap, HashSet};
use std::error::Error;

use rust_htslib::{bam, bam::Read, faidx};

use super::ffi::*;
use super::gpu::{check, Admission, GpuEngine, ReadSoa};
use super::options::CallableOptions;
use super::profilers::{callable_profiler::CallableProfiler, contig_profiler::ContigProfiler};
use super::types::CalledState;

pub fn process_single_contig(bam: &mut bam::IndexedReader, fasta: &mut faidx::Reader,
        header: &bam::HeaderView, counter: &mut CallableProfiler,
        contig_stats: &mut HashMap<usize, ContigProfiler>, options: &CallableOptions,
        tid: usize, gpu: &mut GpuEngine /* owns *mut ClCtx, created once per run */)
        -> Result<(), Box<dyn Error>> {
    let contig_len = header.target_len(tid as u32).unwrap_or(0);
    let contig = std::str::from_utf8(header.tid2name(tid as u32))?;
    bam.fetch((tid as u32, 0, contig_len))?;                       // unchanged, mod.rs:53-54

    // decode: the fields of SURVEY Appendix B
    let mut soa = ReadSoa::default();
    let mut admit = Admission::new(options.max_depth, tid);        // FUNMAP drop + maxcnt rule,
    let mut names = HashSet::new();                                //   = dut_admit_reads (host_coverage.cpp)
    for r in bam.records() {
        let rec = r?;
        if !admit.accept(&rec) { continue; }
        if rec.cigar().end_pos() > rec.pos() { names.insert(rec.qname().to_vec()); }
        soa.push(rec.pos() as i32, rec.mapq(), rec.raw_cigar(), rec.qual());
    }
    // one fetch, not one per base; a zero-length contig (nothing to classify) fetches nothing
    let reference = if contig_len == 0 { Vec::new() } else { fasta.fetch_seq(contig, 0, contig_len as usize - 1)? };

    unsafe {
        check(gpu, cl_contig_begin(gpu.ctx, tid as i32, contig_len as u32, reference.as_ptr(), reference.len() as u64))?;
        check(gpu, cl_push_reads(gpu.ctx, &soa.tile()))?;
        let mut sum = ClContigSummary::default();
        let (mut iv, mut n) = (std::ptr::null(), 0usize);
        check(gpu, cl_contig_finish(gpu.ctx, &mut sum, &mut iv, &mut n))?;
        // CallableProfiler: same state machine, fed run by run instead of position by position
        for run in std::slice::from_raw_parts(iv, n) {
            counter.process_run(contig, run.start as u64, run.end as u64, CalledState::from(run.state))?;
        }
        counter.add_contig_counts(contig, sum.state_counts);
        let st = contig_stats.get_mut(&tid).unwrap();
        st.n_covered_bases = sum.n_covered_bases; st.summed_coverage = sum.summed_coverage;
        st.summed_baseq = sum.summed_baseq; st.summed_mapq = sum.summed_mapq;
        st.quality_bases = sum.quality_bases; st.n_reads = names.len() as u32;
    }
    let stats = &contig_stats[&tid];
    counter.finish_contig(&stats.name, stats.length as u32)?;      // unchanged (keeps the duplicate line)
    Ok(())
}

impl CallableProfiler {
    /// The run-wise form of process_state (callable_profiler.rs:122-155): [start, end) is a maximal stretch of
    /// one state.  Same transitions, `end` taking the place of `pos + 1`; the per-position counts are not touched
    /// here (add_contig_counts adds the device's totals once per contig).
    pub fn process_run(&mut self, contig: &str, start: u64, end: u64, state: CalledState) -> Result<(), Box<dyn Error>> {
        match self.current_state {
            None => {
                // the very first position of the run decides (":128-141"), the rest of the run extends it
                if state != CalledState::REF_N && start > 0 {
                    self.current_state = Some((contig.to_string(), 0, start, CalledState::REF_N));
                    self.write_state()?;
                }
                let s0 = if state == CalledState::REF_N { 0 } else { start };
                self.current_state = Some((contig.to_string(), s0, end, state));
            }
            Some((ref cur_contig, _, ref mut cur_end, ref cur_state)) if cur_contig == contig && *cur_state == state => {
                *cur_end = end;                                  // ":144-146"
            }
            Some(_) => {
                self.write_state()?;                             // ":147-151"
                self.current_state = Some((contig.to_string(), start, end, state));
            }
        }
        Ok(())
    }
    /// contig_counts[contig][state] += n (callable_profiler.rs:124-126), from the device's per-state totals
    pub fn add_contig_counts(&mut self, contig: &str, counts: [u64; 6]) {
        let slot = self.contig_counts.entry(contig.to_string()).or_insert([0; 6]);
        for (i, n) in counts.iter().enumerate() { slot[i] += *n; }
    }
}
