"""Import path kept from the reference (`from lib.pointops2.functions import pointops`,
model/stratified_transformer.py:8): re-exports the MI355X operator API."""
from stratified_transformer_amd.pointops import *  # noqa: F401,F403
from stratified_transformer_amd.pointops import (  # noqa: F401  (names the star-import skips)
    FurthestSampling, KNNQuery, Grouping, AttentionStep1, AttentionStep1_v2, AttentionStep2, AttentionStep2_v2,
    DotProdWithIdx, DotProdWithIdx_v2, DotProdWithIdx_v3, AttentionStep2WithRelPosValue,
    AttentionStep2WithRelPosValue_v2, Subtraction, Aggregation, Interpolation, pointops_cuda)
