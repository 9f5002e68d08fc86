#!/usr/bin/env python3
"""SURVEY.md section 8 "Config 5" (GPU box): 60 s synthetic plucked-string audio (seed-1234 note list, MIDI 40-88, 16 kHz mono)
-> HIP log-mel -> clip windows batched through the model (paper size, random-init weights) -> mpe2note -> MIDI, timed end to end,
scored against the generating note list with evaluation.metrics.

There is no trained checkpoint here (no network): the weights are random, so the note/frame scores printed below say nothing
about transcription accuracy -- they only show that decode + scoring run on real model output.  The throughput numbers are real.
One process = one GPU; with N GPUs the clips of a file are sharded N ways with no collective (replicas only)."""
import json, os, pickle, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import numpy as np
import torch
import bench                                  # paper-size workload table + model builder (no oracle on this path)
from model.amt import AMT
from evaluation.metrics import note_metrics, frame_metrics

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
from corpus import synth_audio as SA
sr, dur, hop = 16000, 60.0, 256
notes = SA.pluck_notes(1234, dur)                   # the seed-1234 note list (corpus/synth_audio.py)
wave = SA.pluck_wave(notes, dur, sr)
config = SA.default_config()
model = bench.build_model(bench.CONFIGS['paper'], 1234, 0.1, 'cpu')
model.hftt_precision = precision
tmp = tempfile.mkdtemp()
with open(os.path.join(tmp, 'model.pkl'), 'wb') as fh:
    pickle.dump(model, fh, protocol=4)
amt = AMT(config, os.path.join(tmp, 'model.pkl'), batch_size=32)

def run():
    t0 = time.time()
    feat = amt.wave2feature(wave.unsqueeze(0), sr)
    torch.cuda.synchronize(); t1 = time.time()
    outs = amt.transcript(feat.numpy())
    torch.cuda.synchronize(); t2 = time.time()
    est = amt.mpe2note(a_onset=outs[4], a_offset=outs[5], a_mpe=outs[6], a_velocity=outs[7])
    t3 = time.time()
    amt.note2midi(est, os.path.join(tmp, 'out.mid'))
    t4 = time.time()
    return feat, outs, est, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)

run()                                               # warm-up (plans, workspaces)
feat, outs, est, (t_fe, t_model, t_dec, t_midi) = run()
n_clips = -(-(feat.shape[0]) // 128)
ref_roll = np.zeros((feat.shape[0], 88), bool)
for n in notes:
    ref_roll[int(n['onset'] * sr / hop):int(n['offset'] * sr / hop), n['pitch'] - 21] = True
line = {'workload': 'config 5: 60 s synthetic audio -> log-mel -> %d clips (paper size, %s mode, random-init weights) -> notes -> MIDI' % (n_clips, precision),
        'frames': int(feat.shape[0]), 'clips': n_clips,
        'seconds': {'logmel': round(t_fe, 4), 'model': round(t_model, 4), 'mpe2note_cpu': round(t_dec, 4), 'note2midi_cpu': round(t_midi, 4)},
        'clips_per_s_model': round(n_clips / t_model, 1), 'clips_per_s_end_to_end': round(n_clips / (t_fe + t_model + t_dec + t_midi), 1),
        'audio_seconds_per_second_end_to_end': round(dur / (t_fe + t_model + t_dec + t_midi), 1),
        'scores_vs_generating_notes_RANDOM_WEIGHTS_no_accuracy_meaning': {
            'note': {k: round(v, 4) for k, v in note_metrics(notes, est).items() if k != 'matching'},
            'frame': {k: round(v, 4) for k, v in frame_metrics(ref_roll, outs[6], threshold=0.5).items()}},
        'n_ref_notes': len(notes), 'n_est_notes': len(est), 'midi_bytes': os.path.getsize(os.path.join(tmp, 'out.mid'))}
print(json.dumps(line))
