"""Stand-in for the reference's global absl ``FLAGS`` object (config/config.py:6-125).

Only the flags the hot path reads are defined (names and defaults as in the reference); modules read
them at construction and at forward time exactly where the reference does (FaceRecon.py:15-16,32-37,
114; PoseNet9D.py:27; PoseR.py:13-14; PoseTs.py:15-16).  Plain attribute bag: ``FLAGS.train = 0``
before constructing a model gives the eval-mode module set, like evaluation/evaluate.py:39.
"""


class _Flags:
    _defaults = dict(
        obj_c=6,              # config.py:6   number of categories
        feat_c_R=1286,        # config.py:31  input channels of the rotation heads
        R_c=4,                # config.py:32
        feat_c_ts=1289,       # config.py:33
        Ts_c=6,               # config.py:34
        feat_face=768,        # config.py:35
        face_recon_c=6 * 5,   # config.py:37
        gcn_sup_num=7,        # config.py:39  support directions S
        gcn_n_num=20,         # config.py:40  neighbours k
        random_points=1028,   # config.py:43
        sample_method='basic',  # config.py:44
        train=1,              # config.py:48
        batch_size=16,        # config.py:55
        lr=1e-4,              # config.py:96
        lr_pose=1.0,          # config.py:98
    )

    def __init__(self):
        self.__dict__.update(self._defaults)

    def reset(self):
        self.__dict__.clear()
        self.__dict__.update(self._defaults)


FLAGS = _Flags()
