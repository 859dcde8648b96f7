"""Shared test helpers (numpy restatements of the reference's MATLAB model used to pin the oracle)."""
import numpy as np


def update_common_signature(mem, dic):
    """archive/2010-LoopClosure/Bayes/updateCommonSignature.m + updateDictionary.m on zero-padded int matrices.

    mem: [n_sig, 1+W] rows (id, words.., 0 pad);  dic: [n_words, 1+R] rows (word id, referencing signature ids.., 0 pad).
    Returns (new virtual-place row, updated dictionary).
    """
    mem = mem.copy()
    dic = dic.copy()
    cs_id = int(mem[0, 0])
    cs = [cs_id]
    # clear references to the virtual place
    for w in mem[0, 1:]:
        idx = np.nonzero(dic[:, 0] == w)[0]
        if w != 0 and idx.size:
            row = dic[idx[0]]
            row[1:][row[1:] == cs_id] = 0
    nb_common = 0
    mem_size = mem.shape[0] - 1
    if mem_size > 0:
        total_active = int(np.count_nonzero(dic[:, 1:]))
        nb_common = total_active // mem_size
    if nb_common > 0:
        counts = np.count_nonzero(dic[:, 1:], axis=1)
        order = np.lexsort((dic[:, 0], counts))          # sortrows([count id])
        lst = np.stack([counts[order], dic[order, 0]], axis=1)
        added = 0
        for i in range(lst.shape[0] - 1, -1, -1):
            if i != lst.shape[0] - 1 and len(cs) > 1:
                ratio = int(lst[i + 1, 0] // lst[i, 0]) if lst[i, 0] else 0
                ln = len(cs)
                stop = False
                for _ in range(2, ratio + 1):
                    for k in range(1, ln):
                        cs.append(cs[k])
                        added += 1
                        if added >= nb_common:
                            break
                    if added >= nb_common:
                        stop = True
                        break
                del stop
            if added < nb_common:
                cs.append(int(lst[i, 1]))
                added += 1
            if added >= nb_common:
                break
        row = np.zeros(mem.shape[1], mem.dtype)
        row[:len(cs)] = cs
        # updateDictionary: first zero slot (or a new column) of each word's row gets the signature id
        for w in cs[1:]:
            if w == 0:
                continue
            idx = np.nonzero(dic[:, 0] == w)[0]
            if not idx.size:
                new = np.zeros((1, dic.shape[1]), dic.dtype)
                new[0, 0] = w
                new[0, 1] = cs_id
                dic = np.vstack([dic, new])
            else:
                r = idx[0]
                zeros = np.nonzero(dic[r] == 0)[0]
                if zeros.size == 0:
                    dic = np.hstack([dic, np.zeros((dic.shape[0], 1), dic.dtype)])
                    dic[r, -1] = cs_id
                else:
                    dic[r, zeros[0]] = cs_id
        return row, dic
    row = np.zeros(mem.shape[1], mem.dtype)
    row[0] = cs_id
    return row, dic


def matlab_compute_likelihood(sign, mem, dic):
    """archive/2010-LoopClosure/Bayes/computeLikelihood.m in float64 (MATLAB doubles)."""
    L = np.zeros(mem.shape[0])
    N = mem.shape[0]
    for w in np.unique(sign[1:]):
        if w == 0:
            continue
        r = np.nonzero(dic[:, 0] == w)[0][0]
        refs = np.unique(dic[r, 1:])
        refs = refs[refs != 0]
        nw = len(refs)
        lg = np.log10(N / nw)
        if lg != 0:
            for s in refs:
                pos = np.nonzero(mem[:, 0] == s)[0][0]
                row = mem[pos]
                nwi = np.count_nonzero(row[1:] == w)
                ni = np.count_nonzero(row[1:] > 0)
                if ni:
                    L[pos] += (nwi * lg) / ni
    return L
