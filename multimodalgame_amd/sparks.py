# coding: utf-8
"""Unicode sparkline used by the training log's message samples (reference: sparks.py:11-14 --
nine glyph levels, values scaled by the maximum of the sequence)."""
GLYPHS = u" ▁▂▃▄▅▆▇▉"


def sparks(nums):
    top = max(nums)
    step = top / float(len(GLYPHS) - 1) if top else 1.0
    return u"".join(GLYPHS[int(round(x / step))] for x in nums)
