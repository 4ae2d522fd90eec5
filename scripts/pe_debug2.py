import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from urmap_amd import api, synth
g = synth.make_genome(101, [180000, 90000, 30000], repeat_frac=0.4, n_families=12)
d = '/tmp/ped'; os.makedirs(d, exist_ok=True)
synth.write_fasta(d + '/small.fa', g, lowercase_frac=0.05)
oi = ol.Index.build(d + '/small.fa', 524309); oi.save(d + '/small.ufi')
rl, s1, s2, indel, n = 120, 0.04, 0.08, 0.01, 2000
r1, r2 = synth.make_pairs(500 + rl, g, n, read_len=rl, sub1=s1, sub2=s2, ins=indel, dele=indel)
seq = oi.seqdata()
idx = api.Index.open(d + '/small.ufi').upload(0); m = api.Mapper(idx)
for pair, (mate, dbpos_other, other_plus) in {719: (1, 287193, True), 1223: (1, 97088, True)}.items():
    read = (r1, r2)[mate][pair][1]
    # other mate plus -> scan(db, 1024, plus=false): query = revcomp(read)
    rc = np.zeros(len(read), np.uint8); ol.lib().uo_revcomp(np.ascontiguousarray(read).ctypes.data, len(read), rc.ctypes.data)
    win = seq[dbpos_other: dbpos_other + 1024].tobytes()
    s, p = ol.viterbi(rc.tobytes(), win, True, True)
    sc, st, paths = m.viterbi_batch([(rc.tobytes(), win)], [3])
    import itertools
    rle = lambda x: ''.join(f"{len(list(gp))}{k}" for k, gp in itertools.groupby(x))
    print(pair, 'oracle', s, rle(p)[-60:], '| gpu', sc[0], st[0], rle(paths[0])[-60:])
