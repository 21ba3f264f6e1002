//! Reference pin: runs the REAL kofft crate on the committed golden inputs and compares bytes with the committed
//! expected outputs (tests/golden/bin, written by tests/golden/export_bin.py from the C restatement under oracle/).
//!
//! The build image of this repository has no Rust toolchain, so this file has never been compiled there: it is the
//! one-command job that turns "parity unpinned" into a reference pin on any box that has cargo:
//!
//!     cd integration/rust/kofft-hip && cargo test --test golden_pin -- --nocapture
//!
//! It needs only the `kofft` dependency (default features, no `+fma`, no SIMD features: the build the published
//! benchmarks use), not the HIP library.  A failure names the case and the first differing element, which says exactly
//! which line of oracle/kofft_oracle_impl.inc misreads which line of kofft.
//!
//! Cases whose expectation depends on how `sin_cos()` lowers (f64 Bluestein chirps, e.g. `c64_15_bluestein`; see
//! tests/test_oracle_second_opinion.py) are reported separately: the oracle takes the merged `sincos` libcall, which is
//! what LLVM emits on x86_64-unknown-linux-gnu.
use kofft::fft::{Complex32, Complex64, FftImpl, FftPlanner, FftStrategy, ScalarFftImpl};
use kofft::rfft::{RealFftImpl, RfftPlanner};
use std::collections::HashMap;
use std::path::PathBuf;

fn dir() -> PathBuf {
    std::env::var_os("KOFFT_GOLDEN_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|| PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../../../tests/golden/bin"))
}

fn bytes(case: &str, field: &str) -> Vec<u8> {
    let p = dir().join(format!("{case}.{field}.bin"));
    std::fs::read(&p).unwrap_or_else(|e| panic!("{}: {e}", p.display()))
}

fn f32s(case: &str, field: &str) -> Vec<f32> {
    bytes(case, field).chunks_exact(4).map(|c| f32::from_le_bytes([c[0], c[1], c[2], c[3]])).collect()
}

fn f64s(case: &str, field: &str) -> Vec<f64> {
    bytes(case, field)
        .chunks_exact(8)
        .map(|c| f64::from_le_bytes([c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]]))
        .collect()
}

fn c32s(case: &str, field: &str) -> Vec<Complex32> {
    f32s(case, field).chunks_exact(2).map(|p| Complex32::new(p[0], p[1])).collect()
}

fn c64s(case: &str, field: &str) -> Vec<Complex64> {
    f64s(case, field).chunks_exact(2).map(|p| Complex64::new(p[0], p[1])).collect()
}

/// The input stream of the `fft_lcg*` cases (tests/golden/export_bin.py: `lcg_values`): every value is an exact f32 in
/// [-1, 1), so a seed in the manifest stands for megabytes of input.
fn lcg_values(seed: u64, count: usize) -> Vec<f32> {
    let mut state = seed;
    (0..count)
        .map(|_| {
            state = state.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407);
            ((state >> 40) & 0xFF_FFFF) as f32 / 8388608.0 - 1.0
        })
        .collect()
}

/// FNV-1a over 8-byte little-endian words (export_bin.py: `fnv1a64`).
fn fnv1a64(data: &[u8]) -> u64 {
    let mut h: u64 = 0xCBF29CE484222325;
    for w in data.chunks_exact(8) {
        h = (h ^ u64::from_le_bytes([w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]])).wrapping_mul(0x100000001B3);
    }
    h
}

/// Bitwise comparison (so that -0.0 != 0.0 and NaN payloads count); returns the first differing index.
fn diff32(got: &[f32], want: &[f32]) -> Option<usize> {
    if got.len() != want.len() {
        return Some(usize::MAX);
    }
    got.iter().zip(want).position(|(a, b)| a.to_bits() != b.to_bits())
}
fn diff64(got: &[f64], want: &[f64]) -> Option<usize> {
    if got.len() != want.len() {
        return Some(usize::MAX);
    }
    got.iter().zip(want).position(|(a, b)| a.to_bits() != b.to_bits())
}
fn flat32(v: &[Complex32]) -> Vec<f32> {
    v.iter().flat_map(|c| [c.re, c.im]).collect()
}
fn flat64(v: &[Complex64]) -> Vec<f64> {
    v.iter().flat_map(|c| [c.re, c.im]).collect()
}

struct Report {
    failures: Vec<String>,
    lowering_dependent: Vec<String>,
    checked: usize,
}

impl Report {
    fn check32(&mut self, case: &str, what: &str, got: &[f32], want: &[f32]) {
        self.checked += 1;
        if let Some(i) = diff32(got, want) {
            let (g, w) = if i < got.len() && i < want.len() { (got[i], want[i]) } else { (f32::NAN, f32::NAN) };
            self.failures.push(format!("{case}/{what}: first difference at element {i}: kofft {g:e} vs oracle {w:e}"));
        }
    }
    fn check64(&mut self, case: &str, what: &str, got: &[f64], want: &[f64]) {
        self.checked += 1;
        if let Some(i) = diff64(got, want) {
            let (g, w) = if i < got.len() && i < want.len() { (got[i], want[i]) } else { (f64::NAN, f64::NAN) };
            let msg = format!("{case}/{what}: first difference at element {i}: kofft {g:e} vs oracle {w:e}");
            if case.starts_with("c64_") && case.ends_with("_bluestein") {
                self.lowering_dependent.push(msg); // f64 chirp tables: sincos vs sin + cos (last bit)
            } else {
                self.failures.push(msg);
            }
        }
    }
}

fn kv(fields: &[&str]) -> HashMap<String, String> {
    fields
        .iter()
        .filter_map(|f| f.split_once('='))
        .map(|(k, v)| (k.to_string(), v.to_string()))
        .collect()
}

#[test]
fn golden_vectors_match_the_reference_bit_for_bit() {
    let manifest = std::fs::read_to_string(dir().join("manifest.tsv")).expect("manifest.tsv");
    let mut rep = Report { failures: vec![], lowering_dependent: vec![], checked: 0 };
    for line in manifest.lines().filter(|l| !l.trim().is_empty()) {
        let cols: Vec<&str> = line.split('\t').collect();
        let (kind, case) = (cols[0], cols[1]);
        let p = kv(&cols[2..]);
        let num = |k: &str| -> usize { p[k].parse().unwrap() };
        match kind {
            "fft" if p["dtype"] == "c32" => {
                let fft = ScalarFftImpl::<f32>::default();
                let mut y = c32s(case, "x");
                fft.fft(&mut y).unwrap();
                rep.check32(case, "fft", &flat32(&y), &f32s(case, "y"));
                let mut z = c32s(case, "x");
                fft.ifft(&mut z).unwrap();
                rep.check32(case, "ifft", &flat32(&z), &f32s(case, "y_inv"));
            }
            "fft" => {
                let fft = ScalarFftImpl::<f64>::default();
                let mut y = c64s(case, "x");
                fft.fft(&mut y).unwrap();
                rep.check64(case, "fft", &flat64(&y), &f64s(case, "y"));
                let mut z = c64s(case, "x");
                fft.ifft(&mut z).unwrap();
                rep.check64(case, "ifft", &flat64(&z), &f64s(case, "y_inv"));
            }
            // one case per dispatch route of the device (two / three factors, the largest single-workgroup size): the input
            // is an LCG stream, the expectation a full spectrum or its hash
            "fft_lcg" | "fft_lcg_hash" => {
                let (n, seed) = (num("n"), p["seed"].parse::<u64>().unwrap());
                let v = lcg_values(seed, 2 * n);
                let out_bytes: Vec<u8> = if p["dtype"] == "c32" {
                    let mut y: Vec<Complex32> = v.chunks_exact(2).map(|q| Complex32::new(q[0], q[1])).collect();
                    ScalarFftImpl::<f32>::default().fft(&mut y).unwrap();
                    if kind == "fft_lcg" {
                        rep.check32(case, "fft", &flat32(&y), &f32s(case, "y"));
                    }
                    flat32(&y).iter().flat_map(|f| f.to_le_bytes()).collect()
                } else {
                    let mut y: Vec<Complex64> = v.chunks_exact(2).map(|q| Complex64::new(q[0] as f64, q[1] as f64)).collect();
                    ScalarFftImpl::<f64>::default().fft(&mut y).unwrap();
                    if kind == "fft_lcg" {
                        rep.check64(case, "fft", &flat64(&y), &f64s(case, "y"));
                    }
                    flat64(&y).iter().flat_map(|f| f.to_le_bytes()).collect()
                };
                if kind == "fft_lcg_hash" {
                    rep.checked += 1;
                    let got = format!("{:016x}", fnv1a64(&out_bytes));
                    if got != p["fnv1a64"] {
                        rep.failures.push(format!("{case}/fft: spectrum hash {got} vs oracle {}", p["fnv1a64"]));
                    }
                }
            }
            // FftStrategy::Radix4 -> fft_radix4 (fft.rs:1356, 1455-1548): the arm kofft_hip_fft_radix4_* reproduces on request
            "radix4" if p["dtype"] == "c32" => {
                let mut y = c32s(case, "x");
                ScalarFftImpl::<f32>::default().fft_with_strategy(&mut y, FftStrategy::Radix4).unwrap();
                rep.check32(case, "fft_with_strategy(Radix4)", &flat32(&y), &f32s(case, "y"));
            }
            "radix4" => {
                let mut y = c64s(case, "x");
                ScalarFftImpl::<f64>::default().fft_with_strategy(&mut y, FftStrategy::Radix4).unwrap();
                rep.check64(case, "fft_with_strategy(Radix4)", &flat64(&y), &f64s(case, "y"));
            }
            "rfft" if p["dtype"] == "f32" => {
                let fft = ScalarFftImpl::<f32>::default();
                let n = num("n");
                let mut x = f32s(case, "x");
                if p["window"] == "1" {
                    // the framing product of stft.rs:96, fused into the batched rfft entry of the C ABI
                    for (v, w) in x.iter_mut().zip(f32s(case, "window")) {
                        *v *= w;
                    }
                }
                let mut y = vec![Complex32::new(0.0, 0.0); n / 2 + 1];
                fft.rfft(&mut x, &mut y).unwrap();
                rep.check32(case, "rfft", &flat32(&y), &f32s(case, "y"));
                let mut spec = c32s(case, "y");
                let mut back = vec![0.0f32; n];
                fft.irfft(&mut spec, &mut back).unwrap();
                rep.check32(case, "irfft", &back, &f32s(case, "x_back"));
            }
            "rfft" => {
                let fft = ScalarFftImpl::<f64>::default();
                let n = num("n");
                let mut x = f64s(case, "x");
                let mut y = vec![Complex64::new(0.0, 0.0); n / 2 + 1];
                fft.rfft(&mut x, &mut y).unwrap();
                rep.check64(case, "rfft", &flat64(&y), &f64s(case, "y"));
                let mut spec = c64s(case, "y");
                let mut back = vec![0.0f64; n];
                fft.irfft(&mut spec, &mut back).unwrap();
                rep.check64(case, "irfft", &back, &f64s(case, "x_back"));
            }
            "stft" => {
                let fft = ScalarFftImpl::<f32>::default();
                let (signal, window) = (f32s(case, "signal"), f32s(case, "window"));
                let mut frames = vec![Vec::new(); num("frames")];
                kofft::stft::stft(&signal, &window, num("hop"), &mut frames, &fft).unwrap();
                let flat: Vec<f32> = frames.iter().flat_map(|f| flat32(f)).collect();
                rep.check32(case, "stft", &flat, &f32s(case, "frames"));
            }
            "istft" => {
                let fft = ScalarFftImpl::<f32>::default();
                let win = num("win");
                let window = f32s(case, "window");
                let mut frames: Vec<Vec<Complex32>> = c32s(case, "frames").chunks(win).map(|c| c.to_vec()).collect();
                let mut output = vec![0.0f32; num("out_len")];
                let mut scratch = vec![0.0f32; num("out_len")];
                kofft::stft::istft(&mut frames, &window, num("hop"), &mut output, &mut scratch, &fft).unwrap();
                rep.check32(case, "istft output", &output, &f32s(case, "output"));
                rep.check32(case, "istft scratch", &scratch, &f32s(case, "scratch"));
            }
            "mags" => {
                let samples = f32s(case, "samples");
                let (mags, max_mag) = kofft::visual::spectrogram::stft_magnitudes(&samples, num("win"), num("hop")).unwrap();
                let flat: Vec<f32> = mags.into_iter().flatten().collect();
                rep.check32(case, "stft_magnitudes", &flat, &f32s(case, "mags"));
                rep.check32(case, "max_mag", &[max_mag], &f32s(case, "max"));
            }
            "twiddles" if p["dtype"] == "f32" => {
                let t = FftPlanner::<f32>::new().get_twiddles(num("n"));
                rep.check32(case, "get_twiddles", &flat32(&t), &f32s(case, "table"));
            }
            "twiddles" => {
                let t = FftPlanner::<f64>::new().get_twiddles(num("n"));
                rep.check64(case, "get_twiddles", &flat64(&t), &f64s(case, "table"));
            }
            "rffttab" if p["dtype"] == "f32" => {
                let t = RfftPlanner::<f32>::new().get_twiddles(num("m"));
                rep.check32(case, "rfft table", &flat32(&t), &f32s(case, "table"));
            }
            "rffttab" => {
                let t = RfftPlanner::<f64>::new().get_twiddles(num("m"));
                rep.check64(case, "rfft table", &flat64(&t), &f64s(case, "table"));
            }
            "hann" => {
                rep.check32(case, "hann", &kofft::window::hann(num("len")), &f32s(case, "table"));
            }
            other => panic!("manifest kind {other} not handled"),
        }
    }
    println!("{} comparisons, {} failed, {} lowering-dependent (f64 Bluestein)", rep.checked, rep.failures.len(),
             rep.lowering_dependent.len());
    for m in &rep.lowering_dependent {
        println!("  lowering-dependent: {m}");
    }
    for m in &rep.failures {
        println!("  FAILED: {m}");
    }
    assert!(rep.checked >= 88, "manifest too short: {}", rep.checked);
    assert!(rep.failures.is_empty(), "{} golden comparisons differ from the reference", rep.failures.len());
}
