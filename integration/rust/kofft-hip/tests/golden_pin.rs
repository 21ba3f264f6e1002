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
use kofft::fft::{Complex32, Complex64, FftImpl, FftPlanner, ScalarFftImpl};
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
    assert!(rep.checked >= 80, "manifest too short: {}", rep.checked);
    assert!(rep.failures.is_empty(), "{} golden comparisons differ from the reference", rep.failures.len());
}
