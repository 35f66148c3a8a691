"""Known answers PUBLISHED by TensorFlow for the ops this path fuses — the worked examples of the TensorFlow API
documentation (r2.6; the same numbers appear in the ops' docstrings in tensorflow/python/ops).  They are the only
vectors for this path that neither this repository nor the reference produced: the reference targets these TF-CPU
semantics (BASELINE.json north_star: "max-abs-diff < 1e-5 vs TF-CPU") and owns no vectors of its own.  Data only:
inputs and the documented outputs."""
import numpy as np

# tf.raw_ops.Bucketize / tf.feature_column.bucketized_column (math_ops bucketize docstring):
#   boundaries = [0, 10, 100], input = [[-5, 10000], [150, 10], [5, 100]] -> [[0, 3], [3, 2], [1, 3]]
BUCKETIZE = dict(boundaries=np.asarray([0, 10, 100], np.float32),
                 values=np.asarray([[-5, 10000], [150, 10], [5, 100]], np.float32),
                 expected=np.asarray([[0, 3], [3, 2], [1, 3]], np.int32))

# tf.gather docstring: params [[0, 1.0, 2.0], [10.0, 11.0, 12.0], [20.0, 21.0, 22.0], [30.0, 31.0, 32.0]], indices [3, 1]
GATHER = dict(params=np.asarray([[0, 1, 2], [10, 11, 12], [20, 21, 22], [30, 31, 32]], np.float32),
              indices=np.asarray([3, 1], np.int64),
              expected=np.asarray([[30, 31, 32], [10, 11, 12]], np.float32))

# tf.sparse.segment_sum docstring: c = [[1,2,3,4], [-1,-2,-3,-4], [5,6,7,8]]
_C = np.asarray([[1, 2, 3, 4], [-1, -2, -3, -4], [5, 6, 7, 8]], np.float32)
SPARSE_SEGMENT_SUM = [
    # (indices, segment_ids, num_segments or None, expected)
    dict(data=_C, indices=[0, 1], segment_ids=[0, 0], num_segments=1, expected=[[0, 0, 0, 0]]),                 # "two rows, one segment"
    dict(data=_C, indices=[0, 1], segment_ids=[0, 1], num_segments=2, expected=[[1, 2, 3, 4], [-1, -2, -3, -4]]),  # "two rows, two segment"
    dict(data=_C, indices=[0, 1], segment_ids=[0, 2], num_segments=4,                                            # "with missing segment ids"
         expected=[[1, 2, 3, 4], [0, 0, 0, 0], [-1, -2, -3, -4], [0, 0, 0, 0]]),
    dict(data=_C, indices=[0, 1, 2], segment_ids=[0, 0, 1], num_segments=2, expected=[[0, 0, 0, 0], [5, 6, 7, 8]]),  # "all rows, two segments"
]

# tf.math.segment_mean docstring (tf.sparse.segment_mean: "like tf.math.segment_mean", rows selected by indices):
#   c = [[1.0,2,3,4], [4,3,2,1], [5,6,7,8]], segment_ids [0, 0, 1] -> [[2.5, 2.5, 2.5, 2.5], [5, 6, 7, 8]]
SPARSE_SEGMENT_MEAN = dict(data=np.asarray([[1, 2, 3, 4], [4, 3, 2, 1], [5, 6, 7, 8]], np.float32), indices=[0, 1, 2],
                           segment_ids=[0, 0, 1], num_segments=2, expected=[[2.5, 2.5, 2.5, 2.5], [5, 6, 7, 8]])

# tf.scatter_nd docstring: indices [[4], [3], [1], [7]], updates [9, 10, 11, 12], shape [8] -> [0, 11, 0, 10, 9, 0, 0, 12]
# (form 3 of the path is ScatterNd(GatherV2(table, ids)) with ascending row ids: the same scatter, pairs in row order)
SCATTER_ND = dict(indices=[4, 3, 1, 7], updates=[9, 10, 11, 12], size=8, expected=[0, 11, 0, 10, 9, 0, 0, 12])

# tf.concat docstring: t1 = [[1,2,3],[4,5,6]], t2 = [[7,8,9],[10,11,12]], axis 1
CONCAT = dict(inputs=[np.asarray([[1, 2, 3], [4, 5, 6]], np.float32), np.asarray([[7, 8, 9], [10, 11, 12]], np.float32)],
              expected=np.asarray([[1, 2, 3, 7, 8, 9], [4, 5, 6, 10, 11, 12]], np.float32))

# tf.strings.to_hash_bucket_fast docstring: (["Hello", "TensorFlow", "2.x"], 3) -> [0, 2, 2]
TO_HASH_BUCKET_FAST = dict(strings=[b"Hello", b"TensorFlow", b"2.x"], num_buckets=3, expected=[0, 2, 2])
