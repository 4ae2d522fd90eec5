import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from urmap_amd import api, synth
stop = int(sys.argv[1])
g = synth.make_genome(101, [180000, 90000, 30000], repeat_frac=0.4, n_families=12)
os.makedirs('/tmp/dbg', exist_ok=True)
synth.write_fasta('/tmp/dbg/small.fa', g, lowercase_frac=0.05)
oi = ol.Index.build('/tmp/dbg/small.fa', 524309); oi.save('/tmp/dbg/small.ufi')
reads = synth.make_reads(1100, g, 300, read_len=150, sub=0.02, ins=0.002, dele=0.002, random_frac=0.03)
offs = np.zeros(len(reads)+1, np.uint64); offs[1:] = np.cumsum([len(r[1]) for r in reads]); bases = np.concatenate([r[1] for r in reads])
idx = api.Index.open('/tmp/dbg/small.ufi').upload(0)
p = api.params_for_method(6); p.xphase4 = 1 | (stop << 8)
m = api.Mapper(idx, params=p)
gres, gops = m.map_se(bases, offs, allow_unsupported=True)
print('stop', stop, 'maxdblo', gres['coord'].max(), gres['seq_index'][:3], np.unique(gres['status']), flush=True)
import os; os._exit(0)
