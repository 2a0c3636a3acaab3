"""Turn one tools/profile_round.sh run (gpurun_out/prof_<tag>/) into the tracked evidence files under profiles/:
  <tag>_kernel_stats.csv                 rocprofv3 --kernel-trace --stats summary of `bench.py --steps 50 --warmup 10`
  <tag>_bench.json                       the bench line of the same round (no profiler attached)
  <tag>_pmc_and_trace_summary.json       per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes + average durations
  <tag>_sq_counters.json                 per-kernel SQ counters (mean per dispatch)
  r05_pmc_traffic.json                   what bench.py reports as roofline.traffic / roofline_attraction_curvature.traffic (tagged there as file-sourced),
                                         with SQ_WAIT_ANY / SQ_WAVE_CYCLES per kernel (_wait_share)
HBM bytes per launch: both counters are in KiB; gfx950's FETCH_SIZE counts exactly half of a WIDE coalesced read (16 B per lane) and other
access widths are uncalibrated (/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).  So the doubling is applied only to
the kernel whose loads are 16 B per lane -- k_nn_wave: float4 localizations, float4 centroids -- and every kernel gets both bounds:
lower = (FETCH_SIZE + WRITE_SIZE) * 1024, upper = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The gather kernels (dword and 12-byte
accesses: k_attract, k_subspace_point_sums, k_prior_directions, k_solve_update, the grid build) report the LOWER bound as their value.
usage: python tools/summarize_round.py <tag>"""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(root, 'profiles')


def short(name):
    n = name.split('(')[0]
    if n.startswith('void '):
        n = n[5:]
    return n.split('<')[0]


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in [max(glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)]:
        for r in csv.DictReader(open(fn)):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


def mean_tail(v, k):
    """mean over the LAST k dispatches (the timed iterations; set-up launches of the same kernel come first)"""
    v = v[-k:] if len(v) > k else v
    return sum(v) / len(v)


newest = lambda pattern: max(glob.glob(pattern, recursive=True), key=os.path.getmtime)      # (a re-run into the same directory leaves the older files behind)
stats = newest(os.path.join(src, 'trace', '**', '*kernel_stats.csv'))
shutil.copy(stats, os.path.join(dst, tag + '_kernel_stats.csv'))
dur = {}
for r in csv.DictReader(open(stats)):
    dur[short(r['Name'])] = dict(calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3, pct=float(r['Percentage']))
# the same kernels restricted to the bench's TIMED region: the `steps` launches before the 10 extra (per-stage) ones
# (the first launches run on the far-from-converged start state and are slower; bench.py's live HIP-event average covers
# exactly the timed launches, so this is the number it must agree with)
trace = newest(os.path.join(src, 'trace', '**', '*kernel_trace.csv'))
per = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    per[short(r['Kernel_Name'])].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
WARM, STEPS = 10, 50
for k, v in per.items():
    v.sort()
    if k in dur and len(v) >= WARM + STEPS + 10:         # (nw_optimize_layout adds a few probe launches after the warm-up)
        t = [d for _, d in v[len(v) - 10 - STEPS:len(v) - 10]]
        dur[k]['avg_us_timed_region'] = sum(t) / len(t) / 1e3
bench = json.loads(open(os.path.join(src, 'bench.log')).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, tag + '_bench.json'), 'w'), indent=1)

WIDE_LOADS = ('k_nn_wave', 'k_nn_fixup')                # kernels whose global loads are 16 B per lane (float4 localizations / centroids)
# every stage a launch of its own (NW_ATTRACT_IN_NN=0) for the per-kernel numbers; the dominant launch -- k_nn_wave as the timed region runs
# it, with the ring half of the prior and the attraction step inside -- from the passes without the knob
fetch, write = counters('fetch_apart'), counters('write_apart')
fetch_f, write_f = counters('fetch_fused'), counters('write_fused')
iters = 10                                             # bench.py --steps 10 in the PMC passes
traffic = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, {}).get('FETCH_SIZE', [0.0])
    w = write.get(k, {}).get('WRITE_SIZE', [0.0])
    per_iter = max(1, round(len(f) / (iters + 10 + 2 * 5)))      # launches per iteration (warm-up 10 + timed 10 + 10 extra)
    fk, wk = mean_tail(f, iters * per_iter), mean_tail(w, iters * per_iter)
    wide = k in WIDE_LOADS
    traffic[k] = dict(FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk, hbm_bytes_per_launch=((2 if wide else 1) * fk + wk) * 1024,
                      hbm_bytes_lower=(fk + wk) * 1024, hbm_bytes_upper=(2 * fk + wk) * 1024,
                      fetch_rule='x2: 16-byte-per-lane loads (guide)' if wide else 'x1: dword / 12-byte gathers, width uncalibrated (upper bound = x2)', dispatches=len(f),
                      avg_us=dur.get(k, {}).get('avg_us'), avg_us_timed_region=dur.get(k, {}).get('avg_us_timed_region'),
                      calls_in_trace=dur.get(k, {}).get('calls'))
grid = ['k_face_centroids', 'k_scan_tile_sums', 'k_scan_bsums', 'k_scan_final', 'k_centroid_scatter']
summary = dict(traffic_raw=traffic,
               note='per-kernel means over the last dispatches of `bench.py --steps 10 --warmup 10` under rocprofv3 --pmc (FETCH_SIZE and '
                    'WRITE_SIZE in separate passes); hbm_bytes_per_launch = (c*FETCH_SIZE + WRITE_SIZE)*1024 with c = 2 only for the kernels whose loads are 16 B per lane (fetch_rule), both bounds given; avg_us from the '
                    '--kernel-trace --stats pass (' + tag + '_kernel_stats.csv)')
json.dump(summary, open(os.path.join(dst, tag + '_pmc_and_trace_summary.json'), 'w'), indent=1)

sq = counters('sq_apart')
sq_f = counters('sq_fused')
json.dump({k: dict({c: sum(v) / len(v) for c, v in d.items()}, dispatches=max(len(v) for v in d.values())) for k, d in sorted(sq.items())},
          open(os.path.join(dst, tag + '_sq_counters.json'), 'w'), indent=1)

out = {k: traffic[k]['hbm_bytes_per_launch'] for k in ('k_nn_wave', 'k_nn_fixup', 'k_attract', 'k_subspace_point_sums', 'k_prior_directions', 'k_solve_update', 'k_reduce_scalars')
       if k in traffic}
if 'k_nn_wave' in traffic:
    out['k_nn_wave_query_and_ring_only'] = traffic['k_nn_wave']['hbm_bytes_per_launch']
    f = fetch_f.get('k_nn_wave', {}).get('FETCH_SIZE', [0.0]); w = write_f.get('k_nn_wave', {}).get('WRITE_SIZE', [0.0])
    # (the fused launch mixes 16-byte-per-lane loads -- the query -- with dword gathers -- the attraction step: x2 for the query's share of the
    # fetches, measured apart, x1 for the rest)
    fq = traffic['k_nn_wave']['FETCH_SIZE_KB']
    # (the bench's last 10 launches are its per-stage pass, which runs the attraction step apart: the TIMED launches are the 10 before them)
    timed = lambda v: (sum(v[-2 * iters:-iters]) / iters) if len(v) >= 2 * iters else mean_tail(v, iters)
    ff, wf = timed(f), timed(w)
    out['k_nn_wave'] = (ff + min(fq, ff) + wf) * 1024
    out['_bounds_fused_k_nn_wave'] = [(ff + wf) * 1024, (2 * ff + wf) * 1024]
out['grid_build'] = sum(traffic[k]['hbm_bytes_per_launch'] * (3 if k.startswith('k_scan') and False else 1) for k in grid if k in traffic)
if 'k_nn_wave' in sq_f and 'SQ_INSTS_VALU' in sq_f['k_nn_wave']:
    vv = sq_f['k_nn_wave']['SQ_INSTS_VALU']
    out['k_nn_wave_valu_wave_instructions'] = (sum(vv[-2 * iters:-iters]) / iters) if len(vv) >= 2 * iters else mean_tail(vv, iters)
if 'k_nn_wave' in sq and 'SQ_INSTS_VALU' in sq['k_nn_wave']:
    out['k_nn_wave_valu_wave_instructions_query_and_ring_only'] = mean_tail(sq['k_nn_wave']['SQ_INSTS_VALU'], iters)
out['_note'] = ('HBM bytes per launch from separate rocprofv3 --pmc passes: (2*FETCH_SIZE + WRITE_SIZE)*1024 for k_nn_wave (16-byte-per-lane loads: gfx950 FETCH_SIZE '
                'counts half of those, MI355X_MICROARCH.md section HBM), (FETCH_SIZE + WRITE_SIZE)*1024 for the gather kernels (other widths are uncalibrated: their x2 upper '
                'bound is in _bounds); mean of the timed iterations of bench.py --steps 10 --warmup 10; profiles/' + tag + '_*')
out['_bounds'] = {k: [traffic[k]['hbm_bytes_lower'], traffic[k]['hbm_bytes_upper']] for k in traffic}
out['_source_tag'] = tag
out['_wait_share'] = {k: (sum(d['SQ_WAIT_ANY']) / max(sum(d['SQ_WAVE_CYCLES']), 1.0)) for k, d in sq.items() if 'SQ_WAIT_ANY' in d and 'SQ_WAVE_CYCLES' in d}
out['_wait_share_note'] = 'SQ_WAIT_ANY / SQ_WAVE_CYCLES per kernel (share of its waves\' cycles spent waiting for anything), same --pmc pass as the SQ counters'
json.dump(out, open(os.path.join(dst, 'r05_pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
for k in ('k_nn_wave', 'k_nn_fixup', 'k_attract', 'k_subspace_point_sums', 'k_reduce_scalars', 'k_prior_directions', 'k_solve_update', 'k_face_centroids', 'k_scan_tile_sums', 'k_scan_final', 'k_centroid_scatter'):
    if k in dur:
        print('%-24s calls %5d avg %8.1f us (timed region %8.1f us)  %5.1f %%' % (k, dur[k]['calls'], dur[k]['avg_us'], dur[k].get('avg_us_timed_region', float('nan')), dur[k]['pct']))
print('bench', bench['ms_per_step'], bench['value'], bench['roofline'])
