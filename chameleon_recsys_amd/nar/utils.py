"""The three I/O helpers of nar_module/nar/utils.py the NAR training path needs (without TensorFlow):
resolve_files (:42-51, sorted glob), chunks (:53-56), get_tf_dtype (:59-68; here -> codec dtype codes)."""
import glob

from .. import _tfrecord


def resolve_files(regex):
    """List of files matching the glob pattern, sorted (utils.py:42-51: tf.train.match_filenames_once + sorted)."""
    return list(sorted(glob.glob(regex)))


def chunks(l, n):
    """Yield successive n-sized chunks from l (utils.py:53-56)."""
    for i in range(0, len(l), n):
        yield l[i:i + n]


def get_tf_dtype(dtype):
    """utils.py:59-68: 'int' -> int64, 'float' -> float32, 'string'/'bytes' -> bytes."""
    if dtype == 'int':
        return _tfrecord.DT_INT64
    elif dtype == 'float':
        return _tfrecord.DT_FLOAT
    elif dtype == 'string' or dtype == 'bytes':
        return _tfrecord.DT_BYTES
    raise Exception('Invalid dtype "{}"'.format(dtype))
