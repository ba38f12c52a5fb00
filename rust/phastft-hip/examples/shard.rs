//! BASELINE configs[4] from a Rust host: 8192 independent f64 FFTs of 2^20 points, 1024 per GPU, one host thread per device,
//! one RCCL all-gather of the 32-byte digests at the end -- the same program as `tests/cpp/shard_host.cpp` (which IS compiled
//! and run by `tests/test_gpu_parity_r5.py`; no Rust toolchain exists in the build image, so this file is source only).
//!
//!     cargo run --release --example shard -- 1024 10
//!
//! Everything device-side goes through `libphastft_hip.so`'s C ABI (`include/phastft_hip.h`); the three RCCL entry points
//! are declared here (link `rccl`, `amdhip64`).  The reference's only parallel construct is `rayon::join`
//! (`src/parallel.rs:13-24`); its planners are plain values usable from any thread (`src/planner.rs:38-39`).
use std::ffi::c_void;
use std::os::raw::{c_int, c_ulonglong};
use std::sync::{Arc, Barrier};
use std::time::Instant;

#[repr(C)]
struct Opaque {
    _private: [u8; 0],
}
type NcclComm = *mut c_void;

#[link(name = "phastft_hip")]
extern "C" {
    fn phast_planner_dit64_new(n: usize, out: *mut *mut Opaque) -> c_int;
    fn phast_planner_dit64_free(p: *mut Opaque);
    fn phast_fft_64_dit_dev(re: *mut f64, im: *mut f64, n: usize, batch: usize, dist: usize, direction: c_int, pl: *const Opaque, stream: *mut c_void) -> c_int;
    fn phast_fill_f64_dev(re: *mut f64, im: *mut f64, n: usize, batch: usize, dist: usize, seed: c_ulonglong, first_id: c_ulonglong, stream: *mut c_void) -> c_int;
    fn phast_digest_f64_dev(re: *const f64, im: *const f64, n: usize, batch: usize, dist: usize, probe: usize, digest: *mut f64, stream: *mut c_void) -> c_int;
}
#[link(name = "amdhip64")]
extern "C" {
    fn hipGetDeviceCount(count: *mut c_int) -> c_int;
    fn hipSetDevice(dev: c_int) -> c_int;
    fn hipMalloc(p: *mut *mut c_void, bytes: usize) -> c_int;
    fn hipFree(p: *mut c_void) -> c_int;
    fn hipDeviceSynchronize() -> c_int;
    fn hipMemcpy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: c_int) -> c_int;
}
#[link(name = "rccl")]
extern "C" {
    fn ncclCommInitAll(comms: *mut NcclComm, ndev: c_int, devlist: *const c_int) -> c_int;
    fn ncclAllGather(send: *const c_void, recv: *mut c_void, count: usize, dtype: c_int, comm: NcclComm, stream: *mut c_void) -> c_int;
    fn ncclCommDestroy(comm: NcclComm) -> c_int;
}
const NCCL_DOUBLE: c_int = 8; // ncclFloat64 (rccl.h: ncclDataType_t)
struct SendPtr<T>(T);
unsafe impl<T> Send for SendPtr<T> {}

fn main() {
    let args: Vec<usize> = std::env::args().skip(1).map(|a| a.parse().expect("shard steps")).collect();
    let (shard, steps) = (*args.first().unwrap_or(&1024), *args.get(1).unwrap_or(&10));
    let n = 1usize << 20;
    let mut gpus: c_int = 0;
    assert_eq!(unsafe { hipGetDeviceCount(&mut gpus) }, 0, "no HIP device: the library has no CPU path");
    let g = gpus as usize;
    let devs: Vec<c_int> = (0..gpus).collect();
    let mut comms: Vec<NcclComm> = vec![std::ptr::null_mut(); g];
    assert_eq!(unsafe { ncclCommInitAll(comms.as_mut_ptr(), gpus, devs.as_ptr()) }, 0);
    let bar = Arc::new(Barrier::new(g));
    let handles: Vec<_> = (0..g)
        .map(|d| {
            let (bar, comm) = (bar.clone(), SendPtr(comms[d]));
            std::thread::spawn(move || unsafe {
                let comm = comm;
                assert_eq!(hipSetDevice(d as c_int), 0);
                let mut pl: *mut Opaque = std::ptr::null_mut();
                assert_eq!(phast_planner_dit64_new(n, &mut pl), 0); // one planner per device
                let (mut re, mut im, mut dig, mut all) = (std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut());
                assert_eq!(hipMalloc(&mut re, shard * n * 8) | hipMalloc(&mut im, shard * n * 8), 0);
                assert_eq!(hipMalloc(&mut dig, shard * 32) | hipMalloc(&mut all, shard * g * 32), 0);
                let (re, im) = (re as *mut f64, im as *mut f64);
                let null = std::ptr::null_mut();
                let first = (d * shard) as c_ulonglong; // transform ids [d * shard, (d + 1) * shard)
                assert_eq!(phast_fill_f64_dev(re, im, n, shard, n, 0xCAFE, first, null), 0);
                assert_eq!(phast_fft_64_dit_dev(re, im, n, shard, n, 1, pl, null), 0); // warm-up: scratch
                assert_eq!(phast_fill_f64_dev(re, im, n, shard, n, 0xCAFE, first, null), 0);
                hipDeviceSynchronize();
                bar.wait();
                let t0 = Instant::now();
                for _ in 0..steps {
                    assert_eq!(phast_fft_64_dit_dev(re, im, n, shard, n, 1, pl, null), 0);
                }
                hipDeviceSynchronize();
                let secs = t0.elapsed().as_secs_f64();
                bar.wait();
                // after the timed region: a fresh step, its digests, the job's only collective
                assert_eq!(phast_fill_f64_dev(re, im, n, shard, n, 0xCAFE, first, null), 0);
                assert_eq!(phast_fft_64_dit_dev(re, im, n, shard, n, 1, pl, null), 0);
                assert_eq!(phast_digest_f64_dev(re, im, n, shard, n, 1, dig as *mut f64, null), 0);
                assert_eq!(ncclAllGather(dig, all, shard * 4, NCCL_DOUBLE, comm.0, null), 0);
                hipDeviceSynchronize();
                let mut host = vec![0f64; if d == 0 { shard * g * 4 } else { 0 }];
                if d == 0 {
                    hipMemcpy(host.as_mut_ptr() as *mut c_void, all, host.len() * 8, 2 /* hipMemcpyDeviceToHost */);
                }
                phast_planner_dit64_free(pl);
                for p in [re as *mut c_void, im as *mut c_void, dig, all] {
                    hipFree(p);
                }
                (secs, host)
            })
        })
        .collect();
    let results: Vec<(f64, Vec<f64>)> = handles.into_iter().map(|h| h.join().unwrap()).collect();
    for c in comms {
        unsafe { ncclCommDestroy(c) };
    }
    let worst = results.iter().map(|r| r.0).fold(0.0, f64::max);
    let finite = results[0].1.iter().all(|x| x.is_finite());
    println!(
        "{{\"metric\": \"GSamples/s f64 forward FFT N=2^20\", \"value\": {:.3}, \"unit\": \"GSamples/s\", \"n_gpus\": {}, \"steps\": {}, \"ms_per_step\": {:.4}, \"scaling\": \"weak\", \"digests_finite\": {}}}",
        (shard * g * n * steps) as f64 / worst / 1e9, g, steps, 1e3 * worst / steps as f64, finite
    );
}
