"""dev: gpurun_out/final6/* (tools/gpu_final_r6.sh on the GPU box) -> profiles/r06_* ; prints the numbers the documents quote."""
import csv, glob, json, os, shutil
O, P = 'gpurun_out/final6', 'profiles'
shutil.copy(O + '/pytest.log', P + '/r06_pytest_gpu.log')
cur = {'_doc': 'per-iteration summaries of every arbiter-checked GPU test of the final tree (tests/arbiter.py): rule by which every body passed, the share within the '
               'LITERAL 1e-4 of the fp32 oracle (no ambiguity slack: bodies_within_1e4_of_oracle32_no_slack / bodies), distances to the fp32 oracle and the fp64 arbiter; '
               'configs2_train_s2_step: the train_s2 step held to an fp64 evaluation per tensor (tests/test_configs2_gpu.py)',
       'tests': {os.path.basename(f)[:-5]: json.load(open(f)) for f in sorted(glob.glob(O + '/arbiter/*.json'))}}
json.dump(cur, open(P + '/r06_arbiter.json', 'w'), indent=1)
for src, dst in (('kernel_stats_sparse_rows.csv', 'r06_kernel_stats_sparse_rows.csv'), ('launch_hist_fwd_scene.txt', 'r06_launch_hist_fwd_scene.txt'),
                 ('timeline_fwd_scene.txt', 'r06_timeline_fwd_scene.txt'), ('timeline_bwd_joint.txt', 'r06_timeline_bwd_joint.txt'), ('sensitivity.json', 'r06_sensitivity.json')):
    if os.path.exists(O + '/' + src):
        shutil.copy(O + '/' + src, P + '/' + dst)
shutil.copy(O + '/kernel_stats.csv', P + '/r06_kernel_stats.csv')
for f in ('r06_pmc_traffic.json', 'r06_pmc_fetch_size.csv', 'r06_pmc_write_size.csv'):
    shutil.copy(O + '/' + f, P + '/' + f)
for f in ('bench_default', 'bench_dp1_nccl', 'bench_n2_gloo', 'bench_habitat'):
    open(P + '/r06_%s.json' % f, 'w').write([l for l in open(O + '/%s.json' % f) if l.startswith('{')][-1])
for mode in ('bf16', 'fp32'):
    src = O + '/train_s2_%s_kernel_stats_unfiltered.csv' % mode
    shutil.copy(src, P + '/r06_train_s2_%s_kernel_stats_unfiltered.csv' % mode)
    rows = list(csv.DictReader(open(src)))
    steps = [int(r['Calls']) for r in rows if 'bwd_joint_kernel' in r['Name'] and 'fit_' not in r['Name']][0]          # one launch per step
    keep = [r for r in rows if int(r['Calls']) >= steps - 2]
    tot = sum(float(r['TotalDurationNs']) for r in keep)
    with open(P + '/r06_train_s2_%s_kernel_stats.csv' % mode, 'w', newline='') as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            w.writerow(dict(r, Percentage='%.2f' % (100 * float(r['TotalDurationNs']) / tot)))
    own = lambda n: 'anonymous namespace' in n or 'psi_' in n or '_GLOBAL__N' in n
    share = lambda pred: sum(float(r['TotalDurationNs']) for r in keep if pred(r['Name'])) / tot
    print(mode, 'steps', steps, 'kernel ms/step %.3f' % (tot / steps / 1e6), 'launches/step %.1f' % (sum(int(r['Calls']) for r in keep) / steps),
          'not ours us/step %.0f' % (sum(float(r['TotalDurationNs']) for r in keep if not own(r['Name'])) / steps / 1e3),
          'library rows', [r['Name'][:30] for r in keep if any(k in r['Name'] for k in ('Cijk', 'igemm', 'multi_tensor', 'ck::', 'SubTensor'))])
    print('  shares: bn/pool %.3f conv %.3f linear %.3f adam %.3f body/scene %.3f aten %.3f' % (
        share(lambda n: 'bn_' in n or 'maxpool' in n), share(lambda n: 'conv' in n or 'stem_' in n), share(lambda n: 'linear_' in n),
        share(lambda n: 'adam' in n.lower()), share(lambda n: any(k in n for k in ('bwd_joint', 'skin_', 'kd_query', 'blend_', 'sdf_', 'head_', 'scene_loss', 'cvae_', 'reduce_partials', 'nn_'))),
        share(lambda n: not own(n))))
d = json.load(open(P + '/r06_bench_default.json'))
print('default', d['value'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step'], 'one call', d['loop_as_one_call']['ms_per_step'], 'frac', d['roofline']['frac'],
      d['roofline']['survey_8d_frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print({k: (v.get('ms_per_step'), v.get('frac'), v.get('value')) for k, v in d['secondary'].items()})
rows = list(csv.DictReader(open(P + '/r06_kernel_stats.csv')))
print([(r['Name'].split('(')[0][-26:], '%.2f' % (float(r['AverageNs']) / 1e3)) for r in rows[:6]], 'sum %.1f' % sum(float(r['AverageNs']) / 1e3 for r in rows[:6]))
for f in ('habitat', 'dp1_nccl', 'n2_gloo'):
    h = json.load(open(P + '/r06_bench_%s.json' % f))
    print(f, h['value'], h['ms_per_step'], (h.get('steady_state') or {}).get('ms_per_step'))
