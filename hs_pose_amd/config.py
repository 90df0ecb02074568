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
        aug_pc_pro=0.2, aug_pc_r=0.2, aug_rt_pro=0.3, aug_bb_pro=0.3, aug_bc_pro=0.3,   # config.py:24-28
        fsnet_loss_type='l1',                                                          # config.py:64
        rot_1_w=8.0, rot_2_w=8.0, rot_regular=4.0, tran_w=8.0, size_w=8.0, recon_w=8.0, r_con_w=1.0,   # config.py:66-72
        recon_n_w=3.0, recon_d_w=3.0, recon_v_w=1.0, recon_s_w=0.3, recon_f_w=1.0,      # config.py:74-78
        recon_bb_r_w=1.0, recon_bb_t_w=1.0, recon_bb_s_w=1.0, recon_bb_self_w=1.0,      # config.py:79-82
        geo_p_w=1.0, geo_s_w=10.0, geo_f_w=0.1,                                         # config.py:87-89
        prop_pm_w=2.0, prop_sym_w=1.0, prop_r_reg_w=1.0,                                # config.py:91-93
        lr=1e-4,              # config.py:96
        lr_pose=1.0,          # config.py:98
    )

    def __init__(self):
        self.__dict__.update(self._defaults)

    def reset(self):
        self.__dict__.clear()
        self.__dict__.update(self._defaults)


FLAGS = _Flags()
