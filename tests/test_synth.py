"""The synthetic generators must reproduce SURVEY.md section 8(d)'s PCM checkpoints."""
from pyflac_amd import synth


def test_config1_hash():
    assert synth.pcm_hash(synth.config1_sine()) == '1803c6f83e993286'


def test_config2_hash():
    assert synth.pcm_hash(synth.config2_stereo16(20.0, 0)) == 'e7bc438b8c7beca8'


def test_config4_hash():
    assert synth.pcm_hash(synth.config4_stereo24(10.0, 1)) == '12c18d749d315e88'
