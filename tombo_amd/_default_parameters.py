"""Constants that are part of the resquiggle contract
(values: /root/reference/tombo/_default_parameters.py:34-97,169-178)."""
RNA_SAMP_TYPE = 'RNA'
DNA_SAMP_TYPE = 'DNA'

STANDARD_MODELS = {DNA_SAMP_TYPE: 'tombo.DNA', RNA_SAMP_TYPE: 'tombo.RNA.180mV'}

# (running_stat_width, min_obs_per_base, raw_min_obs_per_base, mean_obs_per_event)
SEG_PARAMS_TABLE = {RNA_SAMP_TYPE: (12, 6, 2, 15), DNA_SAMP_TYPE: (5, 3, 1, 5)}
# (match_evalue, skip_pen, bandwidth, save_bandwidth, max_half_z_score, band_bound_thresh,
#  start_bw, start_save_bw, start_n_bases)
ALGN_PARAMS_TABLE = {
    RNA_SAMP_TYPE: (6, 4, 500, 1500, 20.0, 50, 1000, 3000, 250),
    DNA_SAMP_TYPE: (4.2, 4.2, 300, 1500, 20.0, 40, 750, 2500, 250)}
SIG_MATCH_THRESH = {RNA_SAMP_TYPE: 2, DNA_SAMP_TYPE: 1.1}
OUTLIER_THRESH = 5.0
EXTRA_SIG_FACTOR = 1.1
MASK_BASES = 50
MASK_FILL_Z_SCORE = -15
DEL_FIX_WINDOW = 2
MAX_DEL_FIX_WINDOW = 10
MAX_RAW_CPTS = 200
MIN_EVENT_TO_SEQ_RATIO = 1.1
USE_RNA_EVENT_SCALE = True
RNA_SCALE_NUM_EVENTS = 10000
RNA_SCALE_MAX_FRAC_EVENTS = 0.75
COLLAPSE_RNA_STALLS = True
COLLAPSE_DNA_STALLS = False
MEAN_STALL_PARAMS = dict(window_size=7 * 50, threshold=40, edge_buffer=100,
                         min_consecutive_obs=200, n_windows=7, mini_window_size=50)
STALL_PARAMS = MEAN_STALL_PARAMS
SHIFT_CHANGE_THRESH = 0.1
SCALE_CHANGE_THRESH = 0.1
MAX_SCALING_ITERS = 3
MAX_POINTS_FOR_THEIL_SEN = 1000
# E|N(0,1)| = sqrt(2/pi): the reference integrates scipy.stats.halfnorm numerically
# (tombo_stats.py:84) and lands on this double bit-for-bit.
HALF_NORM_EXPECTED_VAL = float.fromhex('0x1.9884533d43651p-1')
# per-read statistics (row N4; _default_parameters.py:132-134,158, tombo_stats.py:89-112)
OCLLHR_SCALE = 4.0
OCLLHR_HEIGHT = 1.0
OCLLHR_POWER = 0.2
SMALLEST_PVAL = 1e-50
FM_OFFSET_DEFAULT = 1
SAMP_COMP_TXT = 'sample_compare'
DE_NOVO_TXT = 'de_novo'
ALT_MODEL_TXT = 'model_compare'
CONST_SD_MODEL = True
