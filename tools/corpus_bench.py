#!/usr/bin/env python3
"""Throughput of the device-resident corpus kernels at BASELINE sizes (batch 64 and 512; H=50, T=32, C=128, G=68)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd.corpus import from_synth, history_graph
from nnr_amd.synth import SynthSpec, SynthCorpus

synth = SynthCorpus(SynthSpec(vocabulary_size=60000))
for graph in ('build', 'table'):
    dc = from_synth(synth, 4096, np.random.default_rng(1), 'cuda', graph=graph)
    print('graph=%s: %.1f MB resident (%d news, %d behaviours)' % (graph, dc.resident_bytes() / 1e6, synth.category.shape[0], dc.num))
    for B in (64, 512, 4096):
        idx = torch.randperm(4096, device='cuda')[:B].to(torch.int32)
        for _ in range(3):
            out = dc.train_batch(idx)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            out = dc.train_batch(idx)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        nbytes = sum(t.numel() * t.element_size() for t in out)
        print('  batch %4d: %.3f ms per batch = %.2f M impressions/s; %.1f MB written, %.2f TB/s algorithmic (read + write)' %
              (B, ms, B / ms / 1e3, nbytes / 1e6, 2 * nbytes / ms / 1e9))
cats = torch.randint(0, 18, (4096, 50), device='cuda', dtype=torch.int32)
hm = torch.arange(50, device='cuda')[None, :] < torch.randint(0, 51, (4096, 1), device='cuda')
for _ in range(3):
    history_graph(cats, hm, 18)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    g = history_graph(cats, hm, 18)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print('history_graph_kernel: 4096 graphs of 68x68 in %.3f ms = %.2f TB/s of graph bytes written (%.1f MB)' % (ms, 4096 * 68 * 68 * 4 / ms / 1e9, 4096 * 68 * 68 * 4 / 1e6))
