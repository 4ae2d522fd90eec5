import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from urmap_amd import api, synth
g = synth.make_genome(101, [180000, 90000, 30000], repeat_frac=0.4, n_families=12)
os.makedirs('/tmp/dbg', exist_ok=True)
synth.write_fasta('/tmp/dbg/small.fa', g, lowercase_frac=0.05)
oi = ol.Index.build('/tmp/dbg/small.fa', 524309); oi.save('/tmp/dbg/small.ufi')
reads = synth.make_reads(1100, g, 1500, read_len=100, sub=0.02, ins=0.002, dele=0.002, random_frac=0.03)
offs = np.zeros(len(reads)+1, np.uint64); offs[1:] = np.cumsum([len(r[1]) for r in reads]); bases = np.concatenate([r[1] for r in reads])
ores, opaths, cnt = oi.map_se(bases, offs)
idx = api.Index.open('/tmp/dbg/small.ufi').upload(0); m = api.Mapper(idx)
gres, gops = m.map_se(bases, offs, allow_unsupported=True)
bad = np.nonzero(gres['status'])[0]
print('bad', len(bad), np.unique(gres['status'][bad]))
for i in bad[:5]:
    print(i, gres[i], ores[i])
