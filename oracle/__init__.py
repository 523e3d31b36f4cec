"""ORACLE — test infrastructure only.

CPU restatements of the reference's hot path (plain torch fp32 + one plain-C
operator) used exclusively as the checker by tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg.  The product package
(pmt_learning_for_semantic_segmentation_and_disparity_amd) never imports this.
"""
