"""The chunks of one utterance through the REFERENCE's own driver object, as a few padded batches.

``infer.py:44-67`` (``run_clip``) hands ``Svc.infer`` one chunk at a time, so the model only ever sees B = 1 -- the latency regime.  This is the
three-edit change of INTEGRATION.md section 4 written out: it uses the reference's ``Svc`` instance and the reference's own collate functions
(``infer_tools/infer_tool.py``: ``temporary_dict2processed_input`` :194-259, ``getitem`` :274-292, ``processed_input2batch`` :295-331,
``after_infer`` :170-192) and changes only WHEN the model is called: every non-silent chunk is pre-processed first, the chunks are grouped by
``SvcPipeline.plan_chunks``' cost model, each group goes through ``svc.model(..., infer=True)`` ONCE -- with ``clip_lens`` so that a clip's
padding is the convs' zero padding, exactly as if it had run alone -- and ``after_infer`` + the vocoder run per chunk, in the order given.

    from diffsvc_amd.svc_chunks import infer_chunks
    results = infer_chunks(svc_model, raw_paths, key=key, acc=acc, use_pe=use_pe, use_crepe=use_crepe, thre=thre)   # [(f0_gt, f0_pred, wav), ...]

Equal to ``[svc_model.infer(p, key, acc, ..., seed=seed, first_clip=i) for i, p in enumerate(raw_paths)]`` up to the operand scheme `auto` picks by
call size (tests/ref_infer_driver.py checks the glue with oracle-backed handles; tests/test_gpu_pipeline.py the device path it drives)."""
from io import BytesIO

import numpy as np
import torch


def infer_chunks(svc, wav_fns, key, acc, use_pe=True, use_crepe=True, thre=0.05, seed=None, first_clip=0, batch=True, **kwargs):
    import infer_tools.infer_tool as IT                     # the reference's driver module (on sys.path wherever ``svc`` came from)
    from .pipeline import plan_chunk_groups
    hparams = IT.hparams
    items = []
    for fn in wav_fns:
        name = svc.project_name if isinstance(fn, BytesIO) else fn.split('/')[-1].split('.')[-2]             # Svc.pre :261-267
        temp = svc.temporary_dict2processed_input(name, {'wav_fn': fn, 'spk_id': svc.project_name}, use_crepe, thre)
        items.append(IT.getitem(temp))
    hparams['pndm_speedup'] = acc                             # Svc.pre :269
    n = len(items)
    lengths = [int(it['mel'].shape[0]) for it in items]
    groups = plan_chunk_groups(svc.model.denoise_fn, lengths, acc) if (batch and n > 1) else [[i] for i in range(n)]
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    preds = [None] * n
    for g in groups:
        b = IT.processed_input2batch([items[i] for i in g])
        b['f0'] = b['f0'] + (key / 12)                        # Svc.infer :149-150
        b['f0'][b['f0'] > np.log2(hparams['f0_max'])] = 0
        lens = [lengths[i] for i in g]
        ragged = len(set(lens)) > 1
        extra = dict(kwargs, seed=seed, clip_ids=torch.tensor([first_clip + i for i in g], dtype=torch.int32).cuda())
        if ragged:
            extra.update(clip_lens=torch.tensor(lens, dtype=torch.int32).cuda(), clip_lens_host=lens)
        spk = b.get('spk_embed') if not hparams.get('use_spk_id') else b.get('spk_ids')
        out = svc.model(b['hubert'].cuda(), spk_embed=spk, mel2ph=b['mel2ph'].cuda(), f0=b['f0'].cuda(), uv=b['uv'].cuda(),
                        energy=b['energy'].cuda(), ref_mels=b['mels'].cuda(), infer=True, **extra)
        mel_out = svc.model.out2mel(out['mel_out'])
        f0_gt = IT.denorm_f0(b['f0'], b['uv'], hparams)
        for r, i in enumerate(g):
            m = lengths[i]
            if use_pe:                                        # a clip's own frames only: padding must not reach the extractor's GroupNorm
                f0_pred = svc.pe(out['mel_out'][r:r + 1, :m].contiguous())['f0_denorm_pred'].detach()
            else:
                f0_pred = out.get('f0_denorm')[r:r + 1, :m]
            preds[i] = {'mels': b['mels'][r:r + 1, :m], 'outputs': mel_out[r:r + 1, :m], 'f0_gt': f0_gt[r:r + 1, :m], 'f0_pred': f0_pred}
    # after_infer + vocoder per chunk, in the order the chunks were given (the vocoder plugin draws its noise per call)
    return [svc.after_infer(preds[i], False, wav_fns[i]) for i in range(n)]
