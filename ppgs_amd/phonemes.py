"""The 40 output categories, in channel order (reference ppgs/phonemes.py:10-50).

The last entry is pypar.SILENCE in the reference, whose value is '<silent>'.
"""

PHONEMES = (
    'aa ae ah ao aw ay b ch d dh eh er ey f g hh ih iy jh k l m n ng ow oy '
    'p r s sh t th uh uw v w y z zh').split() + ['<silent>']

PHONEME_TO_INDEX_MAPPING = {phone: i for i, phone in enumerate(PHONEMES)}

assert len(PHONEMES) == 40
