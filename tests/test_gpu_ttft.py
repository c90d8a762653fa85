"""TTFT harness (tools/ttft.py, `bench.py --ttft`): the hipGraph-captured prefill must produce the first tokens the eager stock
module produces on the same spliced embeddings, and the pipeline's pieces must line up (token count, batch)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_ttft_pipeline_graph_equals_eager():
    import ttft
    from transformers import Qwen2Config, Qwen2ForCausalLM
    dev = torch.device("cuda", 0)
    # the real harness with a 2-layer LLM of the 0.5 B width at 256^2 (16 image tokens): both prefill modes
    small = dict(ttft.QWEN2[896], num_hidden_layers=2)
    orig = ttft.QWEN2[896]
    ttft.QWEN2[896] = small
    try:
        g = ttft.measure(2, 256, 896, steps=2, warmup=1, dev=dev, llm_graph=True)
        e = ttft.measure(2, 256, 896, steps=2, warmup=1, dev=dev, llm_graph=False)
    finally:
        ttft.QWEN2[896] = orig
    assert g["prefill_mode"].startswith("hipGraph") and e["prefill_mode"] == "eager"
    assert g["prompt_tokens"] == e["prompt_tokens"] == ttft.PROMPT_BEFORE + ttft.PROMPT_AFTER + 16
    assert g["ttft_ms_median"] > 0 and e["ttft_ms_median"] > 0

    # same weights, same embeddings: graph replay and eager call agree on the first token
    torch.manual_seed(3)
    cfg = Qwen2Config(max_position_embeddings=4096, rope_theta=1e6, rms_norm_eps=1e-6, **small)
    cfg._attn_implementation = "sdpa"
    with torch.device(dev):
        llm = Qwen2ForCausalLM(cfg).to(torch.bfloat16).eval()
    x = torch.randn(2, 40, 896, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        want = llm(inputs_embeds=x, use_cache=True, logits_to_keep=1).logits[:, -1].argmax(-1)
        static_in = torch.zeros_like(x)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            llm(inputs_embeds=static_in, use_cache=True, logits_to_keep=1)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tok = llm(inputs_embeds=static_in, use_cache=True, logits_to_keep=1).logits[:, -1].argmax(-1)
        static_in.copy_(x)
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(tok, want)
