# import stub: a few constants the reference's dataset/eval helpers touch at import time.
INTER_NEAREST = 0
INTER_LINEAR = 1
MORPH_ELLIPSE = 2
