#!/usr/bin/env python3
"""Builds docs/experiments.md: ONE table of every variant that was built or costed, with the file that holds the numbers, what it did and
whether it was adopted.  Rounds 1-4 are read out of profiles/README.md (tables headed `idea | result` = not adopted, `change | before ->
after` = adopted); rounds 5 and 6 are listed below by hand.    python3 tools/make_experiments_index.py"""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R5 = [
    # (area, variant, where the numbers are, result, adopted)
    ("gridding CTF", "preloadCTF's value for envelope-free CTFs as ONE sinusoid: argument reduced in double, float sine polynomial, no atan2 (`d_ctf_pixel_fast`); general formula within 1e-5 of the minCTF threshold", "profiles/r05_a_kernel_stats_onestream.csv", "`k_rf_pack_grid_ctf` 2.30 -> 1.99 ms (now at the HBM rate of its 11.2 GB)", "yes"),
    ("gridding records", "records written by the FFT's row pass with the cheap CTF (`records_from_images` 1)", "DESIGN.md 5 (round 5)", "3.48 ms against 1.47 + 1.99: equal; 3.20 without any CTF evaluation -- the row kernel's store phase, not the CTF, is what costs (2.2 TB/s of stores at 4 workgroups per CU)", "no"),
    ("gridding records", "... with the CTF values of a block's lines evaluated before the store loop", "DESIGN.md 5 (round 5)", "4.62 ms (336 bytes of scratch per lane)", "no"),
    ("matcher S1", "fp32 prefilter as a tile recursion (`k_pm_prefilter_rec2d`: 3 operations per output and axis, 14 warm-up samples)", "profiles/r05_a_kernel_stats_onestream.csv", "1.03 -> 0.52 ms", "yes"),
    ("matcher S1 / S6", "polynomial B-spline weights, fused multiply-adds written out, patch staging row by row (`k_pm_tr_build`, `k_pm_polar_cells`)", "profiles/r05_a_kernel_stats_onestream.csv", "`k_pm_tr_build<float>` 1.42 -> 0.88 ms (224 -> 98 vector instructions per pixel); `k_pm_polar_cells<float>` 1.11 -> 1.06 (not instruction bound)", "yes"),
    ("matcher S2", "workgroup of 2 x 2 / 2 x 4 / 4 x 2 / 4 x 4 waves sharing operand tiles through the L1 (`contract_shape`)", "DESIGN.md 5 (round 5)", "noise gallery 20.3 / 20.7 / 22.3 / 23.3 ms against 20.3 (1 x 4): the L2 was not the bound", "no"),
    ("matcher S2", "bank without a band limit: contraction for the bounds alone, no coefficients stored (`store_cut` 0); S3 contracts the surviving rows", "profiles/r05_a_bench_default.json (`noise_gallery`)", "contraction 23.8 -> 13.2 ms, S3 1.0 -> 3.1 ms; noise gallery 60.7 -> 70.3 k particles/s (the stores and moduli were half the kernel: 12.0 ms without them)", "yes"),
    ("matcher S5", "fp64 prefilter of the re-scored particles as a tile recursion (`k_pm_prefilter_rec64_2d`); `k_pm_pick` one wave per particle", "DESIGN.md 5 (round 5)", "re-score stage 2.1 -> 1.87 ms per step", "yes"),
    ("gridding kernel", "traverse spaces ordered by plane, voxel queue of the previous visit reused for interior visits of the same plane", "DESIGN.md 5 (round 5)", "24.4 -> 24.0 ms per launch on the bench's 4.1 particles per direction (upper bound with every interior visit reusing: 22.4)", "yes"),
    ("gridding kernel", "x-adjacent voxel pairs as items (VERDICT r04 item 3)", "DESIGN.md 5 (round 5), costed from the ISA", "the second voxel's footprint is displaced by (R0.x, R1.x) -- a general 2-D vector -- so the union window is 5 x 5 and its records cannot be shared without per-lane selects (16 taps x 3 v_cndmask); what remains is the set-up (48 -> ~35 per voxel) against half-empty pairs at odd run lengths: > 170 instructions per 64 voxels, not built", "no"),
    ("gridding kernel", "batches across visits (leftover items of visit k in the first batch of visit k + 1)", "DESIGN.md 5 (round 5), costed", "needs two patch buffers per wave: 2 x 60 KB + table 40 + sums 36 > 160 KB at twelve waves; eight waves fit -- not built", "no"),
    ("gridding kernel", "waves per CU: 8 / 12 / 16 (`grid_waves`)", "DESIGN.md 5 (round 5)", "26.3 / 23.6 / 23.9 ms per launch alone; default bench with 8 waves (room for other kernels' registers and LDS on every CU): 45.0 ms per step against 41.2 -- nothing co-runs usefully", "no (12 stays)"),
    ("matcher S3", "list entries dealt to workgroups XCD by XCD (a particle's rows in one L2); 16 instead of 8 rings per step in `d_row_high`", "DESIGN.md 5 (round 5)", "`k_pm_idft_max3` (survivors) 1.73 -> 1.68 ms; 1.75 with 16 rings: the 8.6 GB of operands per step (350 KB per surviving row) at 4.9 TB/s are the bound", "yes / no"),
    ("matcher S2", "bounds-only contraction in one chunk per batch (the chunk size had still been computed for stored coefficients)", "profiles/r05_b_bench_refsnoise.json", "noise gallery 72 -> 76.6 k particles/s (contraction 13.2 -> 10.7 ms alone)", "yes"),
    ("matcher S2", "operands through LDS by LDS-DMA (`global_load_lds_dwordx4`), stages of four quads in two buffers, one barrier per stage (`k_pm_contract_mfma<1, 4, true>`); same products in the same order", "DESIGN.md 5 (round 5)", "two-level contraction (K0 = 28, sixteen quads per frequency) 1.87 -> 1.73 ms; all 399 frequencies 11.7 -> 12.6 ms: adopted for the first, not for the second -- operand delivery is not what holds the matrix pipes at 51-54 %", "yes / no"),
    ("matcher S6", "`k_pm_bestshift_coarse`: the translated particle's taps from LDS-staged bands of 16 + 1 rows of Mimg instead of four global gathers per pixel (bands with the wrap seam or the extrapolated row keep the gathers)", "DESIGN.md 5 (round 5)", "0.95 -> 0.665 ms (bands of 32 rows: 0.70)", "yes"),
    ("gridding front end", "`k_rf_shift_band`: 256-px images shifted band by band, the 19 coefficient rows of a band of 16 staged in LDS (1.2 instead of 7 coefficient loads per pixel); same bits", "DESIGN.md 5 (round 5)", "0.81 -> 0.73 ms: the loads were not the bound (the B-spline weights of every row, evaluated with the reference's expressions, are half of it)", "yes"),
    ("bench", "three, then four warm-up steps (one per distinct batch) instead of one; `value` on batches resident in HBM, the PCIe-inclusive rate beside it", "profiles/r05_c_bench_default.json", "with one warm-up step the first form measured read 3 % low (buffers of data-dependent size still growing inside the clock); a box with a busy host link ran the streamed form at 85 k where the resident one gave 102 k", "yes"),
    ("library", "pinned staging area of the small host arrays in two halves with an event each", "profiles/r05_c_bench_modegrid.json", "`--mode grid` 105 -> 131 k projections/s (the host waited, at every wrap, for the copies it had just queued)", "yes"),
    ("FlexAlign", "rocprofv3 over the two-lane bench (two host threads)", "tools/collect_r05.sh", "hangs (twice, 40 GPU-minutes lost); one lane profiles fine -- the FlexAlign kernel table is collected with `--fa-lanes 1`", "-"),
    ("FlexAlign", "frame prefilter as a tile recursion (`k_fa_prefilter_rec`), eight row groups per warp workgroup", "profiles/r05_a_bench_default.json (`flexalign`)", "`k_fa_prefilter` 0.278 -> `k_fa_prefilter_rec` 0.058 ms per K3 frame (hidden behind the host's spline fit either way), `k_fa_warp_quads` 0.48 -> 0.36 ms per frame; warp + sum 15.5 -> 14.1 ms per movie; 17.1 -> 18.4-18.7 movies/s", "yes"),
    ("FlexAlign global", "pair windows (`k_fa_pairwin_a2`): all rows of a chunk per thread without conditions, the factors of a ky by scalar loads out of a padded table, every complex multiply-add as two `v_pk_fma_f32`", "profiles/r05_g_flexalign_kernel_stats_one_lane.csv", "4.8 -> 2.1 ms per K3 movie (0.82 of the vector issue slots busy)", "yes"),
    ("FlexAlign global", "column pass of the frame transform pruned to the kept rows, two steps as matrix products on the matrix cores (4092 = 124 x 33, 8 of 33 kept; `pruned_columns` 2)", "gpurun A/B, DESIGN.md 5b", "two Bluestein line passes + twiddle + transpose 215 us -> 144 us per frame; 100 us with ten frames per launch (the second step pads 8 rows to a 128-row tile)", "kept as the fallback"),
    ("FlexAlign global", "... as two small DFTs on the vector ALUs (`k_fa_small_dft`: a lane per column, <= 32 outputs in registers, factors by scalar loads, packed multiply-adds; 4092 = 31 x 132, 30 of 132 kept)", "profiles/r05_g_flexalign_kernel_stats_one_lane.csv", "21 us per frame (0.84 ms per movie against 8.6)", "yes"),
    ("FlexAlign global", "row pass of 5760-point rows in one kernel that writes the kept columns only (`k_fft2d_45x128_rows_kept`)", "profiles/r05_g_flexalign_kernel_stats_one_lane.csv", "45-point transforms in registers (`xh_dft45`): 466 registers, one wave per SIMD, 185 us per frame against 170 for the three kernels; as 45 x 24 sums with scalar-loaded factors: 143 us (latency bound at 2.4 waves per SIMD: 46 KB of LDS per row pair); samples five ahead: no change", "yes (143 us)"),
    ("FlexAlign global", "second pass of the pair windows: a lane per window column and group of rows, U staged through LDS", "gpurun A/B", "global alignment 10.3 -> 9.8 ms per movie", "yes"),
    ("FlexAlign local", "patch correlations (`k_fa_patch_corr2`): first pass like `k_fa_pairwin_a2`, U in LDS; second pass one sweep over kx per lane", "profiles/r05_g_flexalign_kernel_stats_one_lane.csv", "9.5 -> 3.9 -> 2.9 ms per movie", "yes"),
    ("FlexAlign warp", "(sx, sy) += (CX, CY) tmp as one packed fused multiply-add on control points stored side by side", "profiles/r05_g_flexalign_kernel_stats_one_lane.csv", "475 -> 299 vector instructions per pixel for the 64 terms, 356 -> 303 us per frame", "yes"),
    ("FlexAlign warp", "what the pixels of a row share taken out of the pixel: G[j] = sum over (layer, row) per image row (a pre-kernel), the pixel sums 4 terms and subtracts the dropped ones, found by 64 wave-uniform tests", "gpurun A/B", "warp stage 12.0 -> 16.7 ms per movie (a scalar load, a compare and a branch per term, in order); inner columns only (16 tests): 12.9", "no"),
    ("FlexAlign warp", "spline weights without selects (the branch is known from the argument's range), weight products two per instruction", "gpurun A/B", "12.0 -> 12.5 ms: the compiler's version was as short; the kernel issues a vector instruction every cycle (counters: `profiles/r05_g_flexalign_pmc_all_kernels.txt`), 468 per row of 64 pixels", "no"),
    ("FlexAlign global", "row kernel, first step: two lines per lane and twelve outputs per wave (half the scalar loads per multiply-add)", "profiles/r05_g_flexalign_kernel_stats_one_lane.csv", "143 -> 115 us per frame", "yes"),
    ("FlexAlign local", "36 instead of 16 patches per batch; the products' next tile fetched under the current products; K in steps of 32", "gpurun A/B", "local alignment 18.5 -> 17.5 ms per movie; products 4.95 + 2.85 -> 4.69 + 2.49 ms; steps of 32: 4.92 + 2.64", "yes / yes / no"),
    ("FlexAlign local", "U of the patch correlations through global scratch instead of LDS (more waves per CU)", "gpurun A/B", "922 -> 1554 us per launch", "no"),
    ("matcher S5", "`k_pm_rescore_row`: a thread owns a shift of both rows (straight and mirrored), one gather of the twiddle for the two sums; same sums", "profiles/r05_g_step_trace.txt", "377 -> 289 us per step", "yes"),
    ("FlexAlign warp", "the 1e-4 cut as an execution mask: `v_cmpx_lt_f32` switches the dropped lanes off for the one packed multiply-add of the term (inline assembly, EXEC restored in the same statement)", "profiles/r05_g_flexalign_kernel_stats_one_lane.csv", "compare + select + multiply-add -> compare + multiply-add: 299 -> 237 vector instructions for the 64 terms, 303 -> 281 us per frame, 27.5-28 -> 29-29.7 movies/s", "yes"),
    ("FlexAlign", "three and four movies in flight", "gpurun A/B", "25.7 / 27.6 movies/s against 27.8 with two", "no"),
]

R6 = [
    ("host side of the programs", "reader threads that enqueue their own H2D copies on their own streams (16 / 32 / 64 readers)", "profiles/experiments/r06_cli_sweeps.txt", "image loop of `xmipp_angular_projection_matching` 75 / 52 / 30 k particles/s: the runtime serialises the threads and the compute thread's launches with them", "no"),
    ("host side of the programs", "persistent reader pool (pread into page-locked 4 MB pieces) + copier threads that alone talk to the runtime; one / two copiers, pieces of 4 / 8 / 16 MB", "profiles/experiments/r06_cli_sweeps.txt", "one copier 152-158 k, two copiers 176 / 168 / 150 k particles/s at 256 px", "yes (two copiers, 4 MB)"),
    ("host side of the programs", "loader threads and their page-locked memory bound to the device's NUMA node (`xh_device_numa_node`)", "profiles/r06_c_ubench_hostfeed.txt", "H2D beside 16 readers on a two-socket host: 21 GB/s unbound, 56 GB/s bound to either node; readers alone 73 -> 97 GB/s on the device's node", "yes"),
    ("host side of the programs", "whole batches page-locked (2 x 1.07 GB) instead of pieces", "profiles/experiments/r06_cli_sweeps.txt", "page-locking costs 0.25 s per GB: 0.52 s of a 1.4 s run", "no (256 MB of pieces)"),
    ("FlexAlign global", "sums of the pruned transforms and the pair windows in blocks of 16 / 32 terms (block sums apart from the total)", "profiles/experiments/r06_flexalign_precision.txt", "K3 parity sample 7.9e-4 -> 3.5e-5 px, but `k_fa_pairwin_a2` 2.16 -> 4.47 ms (twice the accumulators: 3 instead of 5 waves per SIMD); with half the rows per thread 8.3 -> 9.4 ms per movie for the stage", "no"),
    ("FlexAlign global", "the mean of the correlation map (its (0, 0) coefficient) left out of the window sums and of the full inverse transform: bestShift subtracts it again", "profiles/r06_c_flexalign_precision_k3.txt", "K3 parity sample 7.9e-4 -> 2.6e-6 px at round 5's speed (8e-8 through the full inverse transform): the digits were lost to adding small terms to a sum dominated by the mean, not to the length of the sums", "yes"),
    ("gridding kernel", "kernels of the matcher BESIDE the gridding kernel on the same CUs: `grid_waves` 8 leaves 42 KB of LDS and 176 registers per lane, S6's kernels (<= 100 registers, <= 37 KB) fit", "profiles/r06_c_exp_corun.txt, profiles/r06_c_ubench_coresidency.txt", "they do co-reside (microbenchmark and kernel trace) and the work is purely additive: 4 x S6 (16 ms alone) beside a 33.6 ms insert = 50.0 ms; 29.5 + 16 = 45.1 at twelve waves.  Both sides are bound by vector issue; the fp64 repeat kernels and `k_pm_ringdft_mfma` / `k_pm_idft_max3` (225 / 233 registers) do not fit and block their stream until the launch ends", "no"),
    ("gridding kernel", "the 4 x 4 footprint row by row, the records and table entries of the next row requested before this row's multiply-adds (`XgCfg::HALVES` 4, `XgCfg::PIPE`); also two rows at a time, with and without the prefetch, at 12 and 16 waves", "profiles/experiments/r06_ab_grid_pipeline.txt", "23.49 -> 23.15 ms per launch alone (167 -> 133 registers, bit-identical); two rows: 23.3; sixteen waves no better (23.3-24.7)", "yes (row by row + prefetch at twelve waves)"),
    ("gridding kernel", "taps beyond the blob switched off with `v_cmpx` for their two multiply-adds instead of compare + select to the table's zero entry (`-DXG_TAPMASK=1`; possible once the footprint runs row by row: four distances live instead of sixteen)", "profiles/experiments/r06_ab_grid_pipeline.txt", "dense block 155 -> 140 vector instructions, bit-identical, and 4 % SLOWER (22.9 -> 23.9 ms): the select had been sending 53 % of the table reads to one address (a broadcast); unclamped the dead lanes read all over the LDS, and every `v_cmpx` is an EXEC write the next instruction waits for", "no"),
    ("gridding records", "the pack kernels and the projection FFT's row pass leave the record / spectrum cells no tap can reach alone (pixels further than sizeX + 2 r from the origin only ever meet the table's zero entry; record buffer zeroed when allocated)", "gpurun A/B (`--rf-opt skip_far_cells=0`)", "a quarter of the records, their CTF evaluations and bytes: step 38.76 -> 38.39 ms", "yes"),
    ("gridding kernel", "tap rejection without compare + select: (a) clamp of the table index, (b) index pushed beyond the table by max(d - r^2, 0) 2^44 with packed min / max, (c) execution mask per tap (`v_cmpx`)", "DESIGN.md 9 (round 6), costed from the ISA of the dense block (155 vector instructions: 29 v_pk_add, 15 v_pk_mul, 16 v_pk_fma, 16 v_fmac, 18 v_cvt, 17 v_lshl, 16 v_cmp, 16 v_cndmask)", "(a) keeps table entry 9999 for d in (r^2, r^2 + 0.5 / k]: voxel sets differ (round 2 found the same); (b) exact, but four packed instructions per pair of taps = the two it replaces; (c) saves the select (16 of 155) only with the sixteen distances or masks kept live across the LDS latency: 16 more registers at 167 of 168, or 32 scalar registers in a kernel that already spills 80 -- built later in the round, once the footprint ran row by row (four distances live): slower, see the `v_cmpx` line above", "no"),
]


def main():
    src = open(os.path.join(ROOT, "profiles", "README.md")).read().split("\n")
    rows, mode, section = [], None, ""
    for l in src:
        if l.startswith("#"):
            section = l.lstrip("# ").strip(); mode = None; continue
        if re.match(r"^\|\s*idea\s*\|\s*result", l): mode = "no"; continue
        if re.match(r"^\|\s*change\s*\|", l): mode = "yes"; continue
        if not l.startswith("|"):
            if l.strip() == "": mode = mode
            else: mode = None if not l.startswith("|") and mode and not l.strip().startswith("|") and l.strip() else mode
            continue
        if mode and not re.match(r"^\|\s*-", l):
            c = [x.strip() for x in l.strip().strip("|").split("|")]
            if len(c) >= 2:
                rows.append((section, c[0], c[1], mode))
    out = ["# Experiments index", "",
           "Every variant that was built or costed, one line each.  Rounds 1-4 are lifted from the tables of `profiles/README.md` (which keeps the",
           "commands and the longer readings); `profiles/experiments/*` hold the raw A/B records.  Generated by `tools/make_experiments_index.py`.", "",
           "## Round 6", "", "| area | variant | numbers in | result | adopted |", "|---|---|---|---|---|"]
    for a, v, f, r, ad in R6:
        out.append(f"| {a} | {v} | {f} | {r} | {ad} |")
    out += ["", "## Round 5", "", "| area | variant | numbers in | result | adopted |", "|---|---|---|---|---|"]
    for a, v, f, r, ad in R5:
        out.append(f"| {a} | {v} | {f} | {r} | {ad} |")
    out += ["", "## Rounds 1-4 (from profiles/README.md)", "", "| section of profiles/README.md | variant | result | adopted |", "|---|---|---|---|"]
    for s, v, r, ad in rows:
        out.append(f"| {s[:60]} | {v} | {r} | {ad} |")
    os.makedirs(os.path.join(ROOT, "docs"), exist_ok=True)
    open(os.path.join(ROOT, "docs", "experiments.md"), "w").write("\n".join(out) + "\n")
    print(len(R6), "+", len(R5), "+", len(rows), "rows")


if __name__ == "__main__":
    main()
