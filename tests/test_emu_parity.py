"""The compiled gfx950 kernels of libflacgpu.so, executed WITHOUT a GPU on the ISA-level emulator of tests/emu (test infrastructure,
like oracle/), compared with the CPU oracle: the library's own host code launches the instructions hipcc emitted -- inline assembly,
matrix-core autocorrelation, the wave-parallel Rice parser, the packed-history restore chain -- on an interpreter that stands in for
the HIP runtime in a child process.  Not a GPU parity test (tests/test_gpu_*.py, -m gpu, are those; the emulator has no caches, no
memory model and no timing): what it shows is that the code objects in the tree compute the reference's bytes, on every checkout,
also when no GPU is at hand -- and it runs the paths a GPU only takes under rare timing (the join word arriving late, its wait running
out).  Each group is one child process; sizes are chosen so that the whole file takes a minute or two on the build container."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = os.path.join(ROOT, 'tests', 'emu', 'parity_cases.py')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'

pytestmark = pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='the emulator reads the code objects through llvm-objdump')


def _run(*args, env=None, timeout=900):
    e = dict(os.environ)
    e.pop('PYFLAC_AMD_TESTHOOKS', None)
    e.update(env or {})
    p = subprocess.run([sys.executable, CASES] + [str(a) for a in args], env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
    r = json.loads(line[7:])
    assert 'fault' not in r, r['fault']
    return r


@pytest.mark.parametrize('level,seconds,channels,bps', [(5, 0.3, 2, 16), (0, 0.3, 1, 16), (8, 0.2, 2, 24), (3, 0.25, 2, 16)])
def test_drop_in_classes_on_the_emulated_kernels(level, seconds, channels, bps):
    """StreamEncoder's frames equal the oracle's byte for byte; StreamDecoder (16 bit) / the batch decoder (24 bit) return the input."""
    r = _run('dropin', level, seconds, channels, bps)
    assert r['finish'] and r['frames_equal_oracle'] and r['decoded_equals_input'], r


def test_batch_entry_points_and_the_join_through_the_word_in_memory():
    """flacgpu_encode_streams == oracle; flacgpu_decode_stream_dev from the bytes alone == input, twice over (the second call starts
    from the index tables the first one emptied behind its end)."""
    r = _run('batch', 5, 1.0, 4096)
    assert r['encode_equals_oracle'] and r['direct_path'] == 1, r
    assert all(c['equal'] and c['status_max'] == 0 and c['plane_bits'] == 16 and c['generic'] == 0 for c in r['calls']), r


def test_the_join_word_arriving_late_timing_out_and_replaced_by_events_give_the_same_samples():
    """Round 6's join on paths a GPU takes only under rare timing.  With FLACGPU_DEC_DELAY_US (test-hooks build) a wave idles
    in front of the side streams' kernels, so the restore kernel starts first and its workgroups WAIT for the word (join_late_workgroups
    > 0); with FLACGPU_DEC_GATE=2 the word is never raised, the wait runs out, the workgroup writes nothing and the host repeats the call
    with events; FLACGPU_DEC_GATE=0 joins through events from the start.  Same samples every way."""
    base = _run('batch', 5, 0.6, 4096, env={'PYFLAC_AMD_TESTHOOKS': '1', 'FLACGPU_DEC_GATE': '0'})
    late = _run('batch', 5, 0.6, 4096, env={'PYFLAC_AMD_TESTHOOKS': '1', 'FLACGPU_DEC_DELAY_US': '300'})
    dead = _run('batch', 5, 0.6, 4096, env={'PYFLAC_AMD_TESTHOOKS': '1', 'FLACGPU_DEC_GATE': '2'}, timeout=1800)
    for r in (base, late, dead):
        assert r['encode_equals_oracle'] and all(c['equal'] and c['status_max'] == 0 for c in r['calls']), r
    assert [c['sha'] for c in base['calls']] == [c['sha'] for c in late['calls']] == [c['sha'] for c in dead['calls']]
    assert all(c['late'] > 0 for c in late['calls']), late          # the wait did turn
    assert not any(c['late'] for c in base['calls'])                # (events: the word is not looked at)


@pytest.mark.parametrize('first', [0, 40])
def test_seeded_corpus_cases_on_the_emulated_kernels(first):
    """tests/fuzzgen.py seeds (1-8 channels, 8-32 bit, all levels, odd block sizes, ragged tails, limit_min_bitrate): the batch encoder's
    bytes equal the oracle's, the decoder returns the input.  Cases above 60 000 samples are left to the GPU runs."""
    r = _run('fuzz', first, 40)
    assert r['ran'] >= 20 and r['bad'] == [], r


@pytest.mark.parametrize('level', [5, 8])
def test_ragged_blocks_of_true_32_bit_content_on_the_emulated_kernels(level):
    """Round 6: pipe_eval_cand_w32<..., RAG> and the packing kernel's wide form in the ragged geometry -- written and debugged with no
    GPU at hand, on this emulator: the oracle's bytes, and no block left to the generic kernel."""
    r = _run('w32rag', level)
    assert all(c['equal'] and c['redo'] == 0 for c in r['cases']), r


@pytest.mark.parametrize('level,sched', [(5, None), (8, 21)])
def test_frames_of_24_bit_input_packed_at_their_final_place_on_the_emulated_kernels(level, sched):
    """Round 6: the direct packing kernel in its 64-bit forms -- no chunks, no sizes scan, no assembly kernel for
    17..24-bit input either.  Bytes and frame offsets equal the oracle's, the decoder returns the input; once with the waves in the
    emulator's usual order, once in random order and random slices (the frame buffer in LDS is shared by four waves)."""
    r = _run('direct24', level, env={'GFX950EMU_SCHED': str(sched)} if sched else None)
    assert all(c['equal'] and c['offsets'] and c['decoded'] and c['direct'] == 1 and c['redo'] == 0 for c in r['cases']), r


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_batch_round_trip_under_an_adversarial_scheduler(seed):
    """GFX950EMU_SCHED: the emulator runs the resident waves of every slice in random order, each for a random number of instructions,
    and skips some -- orderings between workgroups and between kernels of different streams that its round-robin never produces (and
    a GPU only sometimes): the direct packing path's look-back over the frame sizes of 30 workgroups, the resolve kernel's tickets,
    the fork through the host, the join word.  The bytes and the samples must not depend on it."""
    r = _run('batch', 5, 2.5, 4096, env={'GFX950EMU_SCHED': str(seed)})
    assert r['encode_equals_oracle'] and r['direct_path'] == 1 and r['blocks'] == 30, r
    assert all(c['equal'] and c['status_max'] == 0 for c in r['calls']), r


def test_the_five_configurations_of_the_baseline_on_the_emulated_kernels():
    """BASELINE.json's configs[0..4] at reduced length (tools/emu_configs.py; the full 1024-stream run is profiles/r06_emu_configs.json):
    the passthrough at full size, single-stream encode + decode from the bytes, 24-bit 96 kHz level 8, and a batch of independent
    streams dealt to eight ranks by shard.streams_for_rank -- every stream the oracle's, every decode the source."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'emu_configs.py'), '--streams', '64'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2000:])


def test_under_a_vector_l1_that_is_never_refreshed_the_hand_offs_still_work_and_the_model_has_teeth():
    """GFX950EMU_L1=<n>: the emulator models the per-CU vector L1 as the weakest thing the hardware may do -- a line a plain load brought
    in is served from there, whatever other CUs have stored since, until a kernel starts or a wave of that CU executes buffer_inv sc1;
    sc1 loads and atomics go past it (MI355X_MICROARCH: 'a CU's vector L1 is never refreshed by another CU's stores').  Two CUs for
    all workgroups (much sharing), the adversarial scheduler on top.  (1) The code in the tree: the direct packing path's look-back,
    the fork, the late join -- its acquire fence invalidates the modelled L1 -- give the oracle's bytes and the source's samples,
    direct path taken.  (2) The control: with sc1 loads served from that L1 like plain ones -- what the look-back would see WITHOUT its
    agent-scope loads -- the chain of frame positions breaks under most seeds (direct_path == 2: the encoder notices a stale word
    through its time limit and falls back to the chunk form, bytes still right).  The model detects the defect class it is for."""
    env = {'GFX950EMU_L1': '2', 'PYFLAC_AMD_TESTHOOKS': '1', 'FLACGPU_DEC_DELAY_US': '300'}
    for seed in (1, 2):
        r = _run('batch', 5, 1.5, 4096, env=dict(env, GFX950EMU_SCHED=str(seed)))
        assert r['encode_equals_oracle'] and r['direct_path'] == 1, r
        assert all(c['equal'] and c['status_max'] == 0 and c['late'] > 0 for c in r['calls']), r
    broken = 0
    for seed in (1, 4, 5):
        r = _run('batch', 5, 1.5, 4096, env={'GFX950EMU_L1': '2', 'GFX950EMU_L1_IGNORE_SC1': '1', 'GFX950EMU_SCHED': str(seed)})
        assert r['encode_equals_oracle'], r          # (the fallback keeps the bytes right)
        broken += r['direct_path'] == 2
    assert broken >= 2


def test_both_autocorrelation_kernels_on_the_emulated_runs():
    """Round 6: launches of up to 256 blocks -- every other case of this file -- take fg_pipe_autoc1_kernel (a workgroup a block); the
    headline launch takes fg_pipe_autoc_kernel (a wave a block), which the test-hooks library puts on small launches with
    FLACGPU_AUTOC1=0.  Same bytes: the drop-in classes at level 8 (partial and punched windows) and the seeded corpus cases on the old
    kernel, and the corpus cases once more on the new one with the waves in random order (nine waves meet at a barrier a chunk)."""
    old = {'PYFLAC_AMD_TESTHOOKS': '1', 'FLACGPU_AUTOC1': '0'}
    r = _run('dropin', 8, 0.2, 2, 24, env=old)
    assert r['finish'] and r['frames_equal_oracle'] and r['decoded_equals_input'], r
    r = _run('fuzz', 0, 40, env=old)
    assert r['ran'] >= 20 and r['bad'] == [], r
    r = _run('fuzz', 0, 40, env={'GFX950EMU_SCHED': '17'})
    assert r['ran'] >= 20 and r['bad'] == [], r
    # (levels 6 - 8, very few blocks: the windows side by side is the release choice; one behind the other with FLACGPU_AUTOC1=3)
    r = _run('dropin', 8, 0.2, 2, 24, env={'PYFLAC_AMD_TESTHOOKS': '1', 'FLACGPU_AUTOC1': '3'})
    assert r['finish'] and r['frames_equal_oracle'] and r['decoded_equals_input'], r


@pytest.mark.parametrize('level,bps,enc_max,dec_max', [(5, 16, 25600, 16700), (8, 24, 98500, 25200)])
def test_instructions_a_block_stay_where_the_design_document_says_they_are(level, bps, enc_max, dec_max, tmp_path):
    """tools/emu_counts.py: the VALU wave-instructions the encode and the decode launch execute per block of 4096 stereo samples -- the
    quantity SQ_INSTS_VALU counts on the MI355X (25 441 a block for the level-5 encode in round 5; the emulator's 25 410 for the same
    code).  DESIGN.md section 7.x quotes them; a change that adds instructions to the hot path has to move these bounds knowingly.
    (The tool also checks bytes == oracle and samples == input in the same run.)"""
    out = tmp_path / 'counts.json'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'emu_counts.py'), '--blocks', '16', '--level', str(level), '--bps', str(bps), '--json', str(out)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads(out.read_text())
    assert r['encode']['valu_wave_insts_per_block'] <= enc_max, r['encode']['valu_wave_insts_per_block']
    assert r['decode']['valu_wave_insts_per_block'] <= dec_max, r['decode']['valu_wave_insts_per_block']
    if bps == 24:
        # (round 6: no chunk form left in the configs[3] launch)
        assert not any('assemble' in k or 'scan_sizes' in k for k in r['encode']['kernels']), sorted(r['encode']['kernels'])


def test_the_serial_path_of_a_one_block_call_stays_where_the_design_document_says_it_is(tmp_path):
    """tools/emu_oneblock.py: StreamEncoder.process() with one block is the serial work of one wave per kernel; the emulator counts the
    instructions of every launch's longest wave.  Round 6's fg_pipe_autoc1_kernel against the wave-a-block kernel on
    the same call: 8 375 instead of 11 470 instructions at level 5, 9 127 (its six chains side by side) instead of 43 218 at level 8."""
    out = tmp_path / 'oneblock.json'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'emu_oneblock.py'), '--json', str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads(out.read_text())
    new5, old5 = r['l5_16bit_release'], r['l5_16bit_autoc1_0']
    new8, old8 = r['l8_24bit_release'], r['l8_24bit_autoc1_0']
    assert new5['longest']['pipe_autoc1_kernel'] <= 8500 < 11000 <= old5['longest']['pipe_autoc_kernel']
    assert new8['longest']['pipe_autoc1_kernel'] <= 9300 and new8['longest']['pipe_autoc_fix_kernel'] <= 1200 and old8['longest']['pipe_autoc_kernel'] >= 40000
    assert new5['serial_path'] <= 19400 and new8['serial_path'] <= 43000
