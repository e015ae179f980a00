//! `fugue-gpu`: the MI355X many-chain backend for Fugue's `src/inference` hot path, bound over the C ABI of
//! `libfugue_amd.so` (`include/fugue_amd.h`).
//!
//! UNVERIFIED SOURCE -- never compiled (no Rust toolchain in the build image).
//!
//! * [`ffi`]     -- `extern "C"` declarations of every entry point of `include/fugue_amd.h`.
//! * [`flatten`] -- turns a `Fn() -> Model<A>` into a fixed-structure site program by PROBING it: the model is run
//!                  through a recording [`fugue::runtime::handler::Handler`] at a base assignment and at perturbed ones;
//!                  `Distribution::describe` (the additive patch in `describe.patch`) exposes kind and parameters.
//! * [`backend`] -- `GpuBackend::{hmc_chain, adaptive_mcmc_chain, adaptive_smc}`: the reference's drivers
//!                  (`hmc.rs:566-583`, `mh.rs:921-1014`, `smc.rs:455-581`) over many chains, results as `(A, Trace)`.
pub mod backend;
pub mod ffi;
pub mod flatten;

pub use backend::{ChainDraws, GpuBackend};
pub use flatten::{flatten, FlatProgram};
