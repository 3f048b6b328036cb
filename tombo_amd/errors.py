"""Status code -> reference TomboError message (the message strings are the failure taxonomy;
misspellings included).  Same numbering in include/tombo_amd.h and oracle/tombo_oracle.h."""
from .tombo_helper import TomboError

OK = 0
MESSAGES = {
    1: 'Too much raw signal for mapped sequence',
    2: 'Fewer changepoints found than requested',
    3: 'Read too short for start/end discovery',
    4: 'Genomic mapping too short for start/end discovery',
    5: 'Poor raw to expected signal matching in beginning of read.',
    6: 'Invalid path through read start',
    7: 'Very poor signal quality. Read likely includes open pore.',
    8: ('Read sequence to signal matching starts too far into events for '
        'full adaptive assignment'),
    9: 'Masked z-score contains too few events.',
    10: 'Adaptive signal to seqeunce alignment extended beyond raw signal',
    11: 'Read event to sequence alignment extends beyond bandwidth',
    12: 'Discordant reference and seqeunce lengths.',
    13: 'Not enough raw signal around potential genomic deletion(s)',
    14: 'Read contains too many potential genomic deletions',
    15: 'Invalid segmentation results.',
    16: 'New segments include zero length events',
    17: 'New segments start with negative index',
    18: 'New segments end past raw signal values',
    19: 'Read failed sequence-based signal re-scaling parameter estimation.',
    20: 'Aligned sequence does not match number of segments produced',
    21: 'Must have raw signal in order to complete re-squiggle algorithm',
    22: 'Invalid sequence encountered from genome sequence.',
}
INTERNAL = 100


def message(code):
    return MESSAGES.get(int(code), 'Unexpected error (status %d)' % int(code))


def raise_for_status(code):
    code = int(code)
    if code == OK:
        return
    if code in MESSAGES:
        raise TomboError(MESSAGES[code])
    raise RuntimeError('Unexpected error in resquiggle engine (status %d)' % code)
