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
rng = np.random.default_rng(rl)
acgt = np.frombuffer(b"ACGT", np.uint8)
for k in range(3, n, 29):
    lab, s, q = r2[k]; r2[k] = (lab, acgt[rng.integers(0, 4, size=len(s))], q)
for k in range(5, n, 31):
    lab, s, q = r1[k]; s = s.copy(); s[int(rng.integers(0, len(s)))] = ord("N"); r1[k] = (lab, s, q)
synth.write_fastq(d + '/r1.fq', r1); synth.write_fastq(d + '/r2.fq', r2)
oi.map_file_pe(d + '/r1.fq', d + '/r2.fq', d + '/o.sam', threads=4)
idx = api.Index.open(d + '/small.ufi').upload(0); m = api.Mapper(idx)
labels, bases, offs, quals = api.interleave_pairs(api.read_fastq_arrays(d + '/r1.fq'), api.read_fastq_arrays(d + '/r2.fq'))
res, ops = m.map_pe(bases, offs)
got = (idx.sam_header_sq() + idx.sam_pe(res, ops, labels, bases, offs, quals)).split(b'\n')
want = open(d + '/o.sam', 'rb').read().split(b'\n')
for i in range(len(want)):
    if got[i] != want[i]:
        r = i - 3
        print('read', r, 'pair', r // 2, res[r])
        print(' GPU', b'\t'.join(got[i].split(b'\t')[:9]))
        print(' ORA', b'\t'.join(want[i].split(b'\t')[:9]))
        mate = r ^ 1
        print('  mate GPU', b'\t'.join(got[3 + mate].split(b'\t')[:9]), res[mate])
