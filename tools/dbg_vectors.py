import sys, json, os
sys.path.insert(0, '/root/repo')
import br_amd
from oracle import oracle as O
d = json.load(open('/root/repo/tests/golden/unit_vectors.json'))
only = sys.argv[1] if len(sys.argv) > 1 else None
for v in d['vectors']:
    if v['ignored']: continue
    if only and not v['name'].startswith(only): continue
    s = O.Solid(v['k'])
    for q in v['set_seqs']: s.set_seq(q.encode())
    for q in v['set_kmers']: s.set(O.seq2bit(q.encode()))
    gs = br_amd.Pcon.from_pcon_solid(s.to_bytes())
    m = v['method']
    c = {'one': lambda: br_amd.One(gs, v['confirm']), 'two': lambda: br_amd.Two(gs, v['confirm']), 'graph': lambda: br_amd.Graph(gs),
         'greedy': lambda: br_amd.Greedy(gs, v['max_search'], v['confirm']), 'gap_size': lambda: br_amd.GapSize(gs, v['confirm'])}[m]()
    for a, b in v['cases']:
        print('RUN', v['name'], a, flush=True)
        got = c.correct(a.encode()).decode()
        print('  ->', got, 'OK' if got == b else 'MISMATCH expected ' + b, flush=True)
