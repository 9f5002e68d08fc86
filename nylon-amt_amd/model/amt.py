"""Drop-in replacement of the reference's ``model/amt.py`` (class ``AMT``) on the MI355X HIP path.

Same surface as hftt_code/model/amt.py: ``AMT(config, model_path, batch_size=1, verbose_flag=False)``,
``wav2feature`` (:34), ``transcript`` (:66), ``transcript_stride`` (:121), ``mpe2note`` (:179), ``note2midi`` (:347).
Differences in mechanism, not in results:
  * wav2feature runs the log-mel kernel of libhftt_hip.so (STFT + sparse mel + log on the GPU) instead of torchaudio;
  * transcript/transcript_stride gather ALL clip windows of a file and run them through the model in batches of
    ``batch_size`` clips (the reference stores batch_size and then runs clips one by one, amt.py:31,88);
  * note2midi writes a standard MIDI file directly (pretty_midi is not a dependency).
The model runs on the GPU only; there is no CPU fallback.
"""
import os
import pickle
import struct
import sys

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from hftt_hip._capi import HfttError   # noqa: E402


class AMT():
    def __init__(self, config, model_path, batch_size=1, verbose_flag=False, rank=None, world=None, device=None, gather='host'):
        """rank / world (default: those of torch.distributed when it is initialised, else 0 / 1): the clip batches of a file are dealt
        to the ranks round-robin and each rank runs its share on ITS GPU -- replicas only, NO collective on the data path (the reference
        runs one clip at a time on one device, amt.py:88).  gather='host' (default): every rank copies its own results to the host and
        rank 0 collects them by clip index through a host (gloo) channel, so RANK 0 returns the whole transcription (amt.py:104-113
        stitching order) and writes the files; the other ranks return arrays in which only their own clips are filled.  gather='all': the
        round-1..5 behaviour, an all_gather that leaves the whole result on every rank (only sensible for a few clips).  gather='none':
        no exchange at all (every rank keeps its shard; for callers that shard by FILE with hftt_hip.ddp.shard_indices)."""
        if gather not in ('host', 'all', 'none'):
            raise HfttError("gather must be 'host', 'all' or 'none'")
        self.gather = gather
        self._host_group = None
        if verbose_flag is True:
            print('torch version: ' + torch.__version__)
            print('torch cuda   : ' + str(torch.cuda.is_available()))
        if rank is None or world is None:
            from hftt_hip.ddp import rank_world
            rank, world = rank_world()
        self.rank, self.world = int(rank), int(world)
        if device is not None:
            self.device = device
        elif torch.cuda.is_available():
            self.device = 'cuda:%d' % int(os.environ.get('LOCAL_RANK', 0)) if self.world > 1 and 'LOCAL_RANK' in os.environ else 'cuda'
        else:
            self.device = 'cpu'      # the model itself will refuse to run there (no CPU fallback)

        self.config = config

        if model_path == None:   # noqa: E711  (kept as in the reference)
            self.model = None
        else:
            with open(model_path, 'rb') as f:
                self.model = pickle.load(f)
            self.model = self.model.to(self.device)
            self.model.eval()
            if hasattr(self.model, 'hftt_freeze_weights'):
                self.model.hftt_freeze_weights(True)           # a transcriber never changes its weights: operands are prepared once
            if verbose_flag is True:
                print(self.model)

        self.batch_size = max(1, int(batch_size))
        self._logmel = None

    # ------------------------------------------------------------------ front end (amt.py:34-63)
    def wav2feature(self, f_wav):
        wave, sr = _load_wav(f_wav)                       # [channels, n] float32 in [-1, 1)
        return self.wave2feature(torch.from_numpy(wave), sr)

    def wave2feature(self, wave, sr):
        """wave [channels, n] or [n] float tensor at sample rate sr -> log-mel [n_frames, n_mels] (CPU tensor like the reference)."""
        from hftt_hip import ops
        fe = self.config['feature']
        if not str(self.device).startswith('cuda'):
            raise HfttError('wav2feature runs the HIP log-mel kernel: a ROCm device is required')
        wave = wave.float()
        wave_mono = wave.mean(dim=0) if wave.dim() == 2 else wave        # torch.mean(wave, dim=0), amt.py:56
        if sr != fe['sr']:
            wave_mono = ops.resample(wave_mono.to(self.device), sr, fe['sr'])      # Resample(sr, 16000), amt.py:57-58: hftt_resample (HIP)
        if self._logmel is None:
            if fe['fft_bins'] != fe['window_length']:
                raise HfttError('window_length must equal fft_bins')
            self._logmel = ops.LogMel(self.device, sr=fe['sr'], n_fft=fe['fft_bins'], hop=fe['hop_sample'], n_mels=fe['mel_bins'],
                                      log_offset=fe['log_offset'])
        feat = self._logmel(wave_mono.to(self.device))
        return feat.cpu()

    # ------------------------------------------------------------------ clip windowing + batched inference
    def _run_windows(self, a_input, starts, mode, ablation_flag):
        """a_input [n_in, n_bins] float32 (numpy); one model call per batch of clip windows starting at `starts`."""
        if ablation_flag is True or mode != 'combination':
            raise HfttError("only mode='combination' without ablation is built (the 1-F-D-T model of model_spec2midi.py)")
        cin = self.config['input']
        width = cin['margin_b'] + cin['num_frame'] + cin['margin_f']
        x = torch.from_numpy(a_input).to(self.device)                          # the whole file's features: one host->device copy
        self.model.eval()
        batches = [starts[b0:b0 + self.batch_size] for b0 in range(0, len(starts), self.batch_size)]
        mine = list(range(self.rank, len(batches), self.world))                 # this rank's batches (round-robin)
        parts = [[] for _ in range(8)]
        for bi in mine:
            spec = torch.stack([x[i:i + width].T for i in batches[bi]], dim=0)  # [b, n_bins, width] (amt.py:89)
            with torch.no_grad():
                o = self.model(spec)
            sel = (o[0], o[1], o[2], o[3].argmax(3), o[5], o[6], o[7], o[8].argmax(3))   # velocity argmax (amt.py:107,113)
            for lst, t in zip(parts, sel):
                lst.append(t)                                                   # stays on the device: one copy back per file, not per batch
        return self._collect(parts, batches, mine)

    def _collect(self, parts, batches, mine):
        """per-rank device results -> 8 numpy arrays [n_clips, num_frame, num_note] in clip order (complete on rank 0; see __init__: gather)"""
        cin, cm = self.config['input'], self.config['midi']
        T, N = cin['num_frame'], cm['num_note']
        n_clips = sum(len(b) for b in batches)
        if self.world == 1:
            return [torch.cat(l, dim=0).to('cpu').numpy() for l in parts]
        import torch.distributed as dist
        if self.gather != 'all':
            # replicas only: each rank's results go to ITS host; rank 0 collects the shards by clip index over a host channel (gloo),
            # never over RCCL -- no device collective anywhere on the inference path
            my_clips = [i for bi in mine for i in range(bi * self.batch_size, bi * self.batch_size + len(batches[bi]))]
            local = [torch.cat(l, dim=0).to('cpu').numpy() if l else np.zeros((0, T, N), np.int64 if k % 4 == 3 else np.float32) for k, l in enumerate(parts)]
            shards = [(my_clips, local)]
            if self.gather == 'host':
                if dist.get_backend() == 'gloo':
                    grp = None
                else:
                    if self._host_group is None:
                        self._host_group = dist.new_group(backend='gloo')      # (collective over all ranks: every rank of the job runs the same AMT calls)
                    grp = self._host_group
                got = [None] * self.world if self.rank == 0 else None
                dist.gather_object((my_clips, local), got, dst=0, group=grp)
                if self.rank == 0:
                    shards = got
            res = []
            for k in range(8):
                full = np.zeros((n_clips, T, N), np.int64 if k % 4 == 3 else np.float32)
                for clips, arrs in shards:
                    if len(clips):
                        full[np.asarray(clips)] = arrs[k]
                res.append(full)
            return res
        gdev = 'cpu' if dist.get_backend() == 'gloo' else self.device          # gloo moves host memory; RCCL ('nccl') device memory
        per = -(-len(batches) // self.world) * self.batch_size                # clips per rank, padded: equal-sized all_gather
        res = []
        order = [i for r in range(self.world) for bi in range(r, len(batches), self.world) for i in range(bi * self.batch_size, bi * self.batch_size + len(batches[bi]))]
        counts = [sum(len(batches[bi]) for bi in range(r, len(batches), self.world)) for r in range(self.world)]
        for k, l in enumerate(parts):
            dt = torch.int64 if k % 4 == 3 else torch.float32
            local = torch.zeros(per, T, N, dtype=dt, device=gdev)
            if l:
                cat = torch.cat(l, dim=0)
                local[:cat.shape[0]] = cat
            gathered = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(gathered, local)
            flat = torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)
            full = torch.empty(n_clips, T, N, dtype=dt, device=gdev)
            full[torch.tensor(order, device=gdev)] = flat
            res.append(full.to('cpu').numpy())
        return res

    def transcript(self, a_feature, mode='combination', ablation_flag=False):
        # a_feature: [num_frame, n_mels]
        a_feature = np.array(a_feature, dtype=np.float32)
        cin, cf, cm = self.config['input'], self.config['feature'], self.config['midi']
        T = cin['num_frame']
        n = a_feature.shape[0]
        a_tmp_b = np.full([cin['margin_b'], cf['n_bins']], cin['min_value'], dtype=np.float32)
        len_s = int(np.ceil(n / T) * T) - n
        a_tmp_f = np.full([len_s + cin['margin_f'], cf['n_bins']], cin['min_value'], dtype=np.float32)
        a_input = np.concatenate([a_tmp_b, a_feature, a_tmp_f], axis=0)
        starts = list(range(0, n, T))
        res = self._run_windows(a_input, starts, mode, ablation_flag)
        outs = []
        for k, r in enumerate(res):
            dt = np.int8 if k % 4 == 3 else np.float32
            full = np.zeros((n + len_s, cm['num_note']), dtype=dt)
            for c, i in enumerate(starts):
                full[i:i + T] = r[c]
            outs.append(full)
        return tuple(outs)

    def transcript_stride(self, a_feature, n_offset, mode='combination', ablation_flag=False):
        # a_feature: [num_frame, n_mels]
        a_feature = np.array(a_feature, dtype=np.float32)
        cin, cf, cm = self.config['input'], self.config['feature'], self.config['midi']
        half_frame = int(cin['num_frame'] / 2)
        n = a_feature.shape[0]
        a_tmp_b = np.full([cin['margin_b'] + n_offset, cf['n_bins']], cin['min_value'], dtype=np.float32)
        tmp_len = n + cin['margin_b'] + cin['margin_f'] + half_frame
        len_s = int(np.ceil(tmp_len / half_frame) * half_frame) - tmp_len
        a_tmp_f = np.full([len_s + cin['margin_f'] + (half_frame - n_offset), cf['n_bins']], cin['min_value'], dtype=np.float32)
        a_input = np.concatenate([a_tmp_b, a_feature, a_tmp_f], axis=0)
        starts = list(range(0, n, half_frame))
        res = self._run_windows(a_input, starts, mode, ablation_flag)
        outs = []
        for k, r in enumerate(res):
            dt = np.int8 if k % 4 == 3 else np.float32
            full = np.zeros((n + len_s, cm['num_note']), dtype=dt)
            for c, i in enumerate(starts):
                full[i:i + half_frame] = r[c][n_offset:n_offset + half_frame]
            outs.append(full)
        return tuple(outs)

    # ------------------------------------------------------------------ posteriorgram -> notes (amt.py:179-344)
    def mpe2note(self, a_onset=None, a_offset=None, a_mpe=None, a_velocity=None, thred_onset=0.5, thred_offset=0.5, thred_mpe=0.5,
                 mode_velocity='ignore_zero', mode_offset='shorter'):
        hop_sec = float(self.config['feature']['hop_sample'] / self.config['feature']['sr'])
        a_onset = np.asarray(a_onset); a_offset = np.asarray(a_offset); a_mpe = np.asarray(a_mpe); a_velocity = np.asarray(a_velocity)
        a_note = []
        n_mpe = len(a_mpe)
        for j in range(self.config['midi']['num_note']):
            on_loc, on_time = _pick_peaks(a_onset[:, j], thred_onset, hop_sec)          # :193-223
            off_loc, off_time = _pick_peaks(a_offset[:, j], thred_offset, hop_sec)      # :224-253
            below = np.nonzero(a_mpe[:, j] < thred_mpe)[0]
            time_offset = 0.0
            for idx_on in range(len(on_loc)):
                loc_onset, time_onset = int(on_loc[idx_on]), float(on_time[idx_on])
                if idx_on + 1 < len(on_loc):                                            # :263-269
                    loc_next, time_next = int(on_loc[idx_on + 1]), float(on_time[idx_on + 1])
                else:
                    loc_next, time_next = n_mpe, (n_mpe - 1) * hop_sec
                loc_offset, flag_offset = loc_onset + 1, False                         # :272-281
                k = np.searchsorted(off_loc, loc_onset, side='right')
                if k < len(off_loc):
                    loc_offset, time_offset, flag_offset = int(off_loc[k]), float(off_time[k]), True
                if loc_offset > loc_next:                                               # :282-284
                    loc_offset, time_offset = loc_next, time_next
                loc_mpe, flag_mpe, time_mpe = loc_onset + 1, False, 0.0                 # :288-296 ("1 frame longer")
                kb = np.searchsorted(below, loc_onset + 1, side='left')
                if kb < len(below) and below[kb] < loc_next:
                    loc_mpe, flag_mpe = int(below[kb]), True
                    time_mpe = loc_mpe * hop_sec
                pitch_value = int(j + self.config['midi']['note_min'])
                velocity_value = int(a_velocity[loc_onset][j])
                if (not flag_offset) and (not flag_mpe):                                # :311-331
                    offset_value = float(time_next)
                elif flag_offset and (not flag_mpe):
                    offset_value = float(time_offset)
                elif (not flag_offset) and flag_mpe:
                    offset_value = float(time_mpe)
                elif mode_offset == 'offset':
                    offset_value = float(time_offset)
                elif mode_offset == 'longer':
                    offset_value = float(time_offset) if loc_offset >= loc_mpe else float(time_mpe)
                else:
                    offset_value = float(time_offset) if loc_offset <= loc_mpe else float(time_mpe)
                if mode_velocity != 'ignore_zero' or velocity_value > 0:                # :332-336
                    a_note.append({'pitch': pitch_value, 'onset': float(time_onset), 'offset': offset_value, 'velocity': velocity_value})
                if len(a_note) > 1 and a_note[-1]['pitch'] == a_note[-2]['pitch'] and a_note[-1]['onset'] < a_note[-2]['offset']:
                    a_note[-2]['offset'] = a_note[-1]['onset']                         # :338-341
        return sorted(sorted(a_note, key=lambda x: x['pitch']), key=lambda x: x['onset'])   # :343

    def note2midi(self, a_note, f_midi):
        """Single-track standard MIDI file, 220 ticks per beat at 120 bpm (pretty_midi's defaults), program 0."""
        ticks_per_sec = 220 * 2.0
        events = []
        for note in a_note:
            on, off = int(round(note['onset'] * ticks_per_sec)), int(round(note['offset'] * ticks_per_sec))
            events.append((on, 1, bytes([0x90, int(note['pitch']) & 0x7F, int(note['velocity']) & 0x7F])))
            events.append((max(off, on), 0, bytes([0x90, int(note['pitch']) & 0x7F, 0])))
        events.sort(key=lambda e: (e[0], e[1]))
        trk = bytearray()
        trk += b'\x00\xff\x51\x03' + (500000).to_bytes(3, 'big')          # tempo 120 bpm
        trk += b'\x00\xff\x58\x04\x04\x02\x18\x08'                         # 4/4
        trk += b'\x00\xc0\x00'                                             # program 0
        t = 0
        for tick, _, msg in events:
            trk += _vlq(tick - t) + msg
            t = tick
        trk += b'\x01\xff\x2f\x00'
        with open(f_midi, 'wb') as f:
            f.write(b'MThd' + struct.pack('>IHHH', 6, 0, 1, 220))
            f.write(b'MTrk' + struct.pack('>I', len(trk)) + bytes(trk))
        return


# ---------------------------------------------------------------------- helpers
def _pick_peaks(a, thred, hop_sec):
    """Local maxima >= thred of one pitch track with the reference's plateau rule (amt.py:195-223): a frame counts when
    the nearest DIFFERENT value on each side is smaller (or there is none); time refined from the two neighbours."""
    a = np.asarray(a, dtype=np.float32)
    n = len(a)
    if n == 0:
        return np.zeros(0, np.int64), np.zeros(0)
    change = np.nonzero(a[1:] != a[:-1])[0] + 1
    run_start = np.concatenate([[0], change])
    run_id = np.zeros(n, np.int64)
    run_id[change] = 1
    run_id = np.cumsum(run_id)
    run_val = a[run_start]
    prev_val = np.concatenate([[-np.inf], run_val[:-1]])[run_id]
    next_val = np.concatenate([run_val[1:], [-np.inf]])[run_id]
    loc = np.nonzero((a >= thred) & (a > prev_val) & (a > next_val))[0]
    times = loc * hop_sec
    # Sub-frame refinement.  The reference mixes Python floats with numpy float32 scalars (amt.py:217-219); under NumPy 2
    # promotion rules (the build container, where the goldens were made) that arithmetic is carried out in float32, under
    # NumPy 1.x in float64 -- a <= 1e-6 s difference.  float32 is restated here so note order ties resolve identically.
    h2 = np.float32(hop_sec * 0.5)
    for q, i in enumerate(loc):
        if i == 0 or i == n - 1:
            continue
        l, c, r = a[i - 1], a[i], a[i + 1]
        if l > r:
            times[q] = float(np.float32(i * hop_sec) - (h2 * (l - r) / (c - r)))
        elif l < r:
            times[q] = float(np.float32(i * hop_sec) + (h2 * (r - l) / (c - l)))
    return loc, times


def _vlq(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    return bytes(reversed(out))


def _load_wav(path):
    """PCM / float WAV -> float32 [channels, n] scaled like torchaudio.load (integers / 2^(bits-1))."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return np.ascontiguousarray(x.T), int(sr)
