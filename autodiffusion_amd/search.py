"""Evolutionary search over timestep subsets -- the reference's driver on the HIP evaluation path.

Drop-in mirror of ``EvolutionSearcher`` (reference search_imagenet64_classifier_guidance.py:155-584;
unconditional variant search_uncondition_model.py; dict candidates with per-step skip lists are
accepted by ``get_cand_fid`` as in search_dynamic_unet_imagenet64_classifier_guidance_progressive.py
:369-445).  The EA operators consume ``random`` / ``np.random`` in exactly the reference's order
(pinned by tests/golden/ea_trajectory.npz), the log lines keep the reference's text, and
``get_cand_fid(cand, args) -> float`` keeps its signature and side effects.

What changes underneath:
  * sampling runs on the HIP engine (evaluate.CandidateEvaluator); images never leave the GPU;
  * FID statistics are accumulated on the GPU and pooled over ranks with one all-gather per
    candidate (fid.ActivationAccumulator) instead of all-gathering every uint8 batch;
  * every batch is seeded by (seed, candidate, batch index, rank), so a candidate's FID does not depend
    on how the work was sharded;
  * the Inception feature extractor is a plug: ``features(uint8 NHWC device batch) -> [B, D]`` -- the bundled HIP
    Inception-v3 pool3 (``inception.InceptionV3(...).features``, parity unpinned: DESIGN.md section 9) or any other callable --
    or an ``Evaluator_v1``-style object (then the reference's host cal_fid path is used).
"""
from __future__ import annotations

import random
import time
import zlib

import numpy as np
import torch

from . import dist_util, logger
from .evaluate import CandidateEvaluator, merge_policy
from .fid import ActivationAccumulator, FIDStatistics, cal_fid
from .schedule import space_timesteps

choice = lambda x: x[np.random.randint(len(x))] if isinstance(x, tuple) else choice(tuple(x))  # noqa: E731


class EvolutionSearcher(object):

    def __init__(self, args, model, base_diffusion, time_step, classifier=None, search_space=None,
                 evaluator=None, ref_stats=None, features=None, feature_dim=2048, variant=None,
                 population_parallel=False):
        self.args = args
        self.model = model
        self.base_diffusion = base_diffusion
        self.classifier = classifier
        self.time_step = time_step
        self.max_epochs = args.max_epochs
        self.select_num = args.select_num
        self.population_num = args.population_num
        self.m_prob = args.m_prob
        self.crossover_num = args.crossover_num
        self.mutation_num = args.mutation_num
        self.keep_top_k = {self.select_num: [], 50: []}
        self.epoch = 0
        self.candidates = []
        self.vis_dict = {}
        self.max_fid = getattr(args, "max_fid", 48.0)
        self.thres = getattr(args, "thres", 0.2)
        self.search_space = search_space
        self.x0 = getattr(args, "init_x", "")
        # "guided" = search_imagenet64_classifier_guidance.py, "unconditional" = search_uncondition_model.py
        # (the latter seeds pop//2 random candidates instead of pop//2 + 1 and stops before the last
        # epoch's offspring, :509-532, :554-555)
        self.variant = variant or ("guided" if classifier is not None else "unconditional")
        self.break_at_last_epoch = self.variant == "unconditional"
        # FID plumbing: reference-style evaluator object, or a device feature function
        self.evaluator = evaluator
        self.features = features
        self.feature_dim = feature_dim
        self.ref_stats = ref_stats
        # an extractor that runs on random weights (inception.pool3_features(allow_random=True): throughput runs, tests) ranks
        # candidates on a meaningless metric: every FID line of log.txt says so, next to the value
        self.fid_note = (" [FID on RANDOM Inception weights: not a quality metric]"
                         if getattr(features, "random_weights", False) else "")
        self.last_times = None   # {'reset_time', 'sample_time', 'fid_time', 'images'} of the last get_cand_fid
        if ref_stats is None and getattr(args, "ref_path", ""):
            # the reference pickles a FIDStatistics; this build reads the two arrays from an .npz
            # (mu, sigma) written from it -- unpickling foreign files is not done here
            z = np.load(args.ref_path, allow_pickle=False)
            self.ref_stats = FIDStatistics(z["mu"], z["sigma"])
        # population parallelism: within an epoch candidate GENERATION never looks at
        # FID values (legality is only the visited-set dedupe; parents are the top-k frozen at epoch start),
        # so candidates are generated first -- with the reference's exact random / np.random draw order, on
        # every rank identically -- queued, then evaluated rank r -> candidates r, r+W, ... (whole-candidate
        # FID local to one GPU) and the FIDs are all-gathered back in order.
        self.population_parallel = population_parallel
        self._pending = []
        self.last_flush = None
        self._ref_dev = None     # (ref_stats, mu, sigma on the device): uploaded once, not once per candidate
        self._ev = None
        if model is not None:
            self._ev = CandidateEvaluator(
                model, base_diffusion, classifier, image_size=args.image_size, use_ddim=args.use_ddim,
                clip_denoised=args.clip_denoised, class_cond=args.class_cond,
                classifier_scale=getattr(args, "classifier_scale", 1.0), device=dist_util.dev(),
                use_graph=bool(getattr(args, "use_graph", False)))
            self.active_diffusion = self._ev.active_diffusion

    # ------------------------------------------------------------------ evaluate-candidate interface
    def reset_diffusion(self, use_timesteps):
        self._ev.set_candidate(list(use_timesteps))

    def get_cand_fid(self, cand=None, args=None):
        if self.population_parallel:
            # whole candidate on this rank alone: its batches are seeded as a single-rank run would seed them, the
            # statistics stay local and there is no collective inside (ranks evaluate different candidates)
            return self._cand_fid(cand, args, world=1, rank=0, local=True)
        return self._cand_fid(cand, args, world=dist_util.get_world_size(), rank=dist_util.get_rank(), local=False)

    def _cand_fid(self, cand, args, *, world, rank, local):
        """`world` / `rank` = how this candidate's image batches are sharded (1 / 0: all of them here)."""
        args = args if args is not None else self.args
        t1 = time.time()
        self._ev.set_candidate(cand)
        reset_time = time.time() - t1
        t1 = time.time()
        logger.log("sampling...")
        seed0 = (int(getattr(args, "seed", 0)) * 1000003 + zlib.crc32(str(cand).encode())) & 0x7FFFFFFF
        # fewer samples than feature dimensions (the search's regime, `num_samples <= 1000` against 2048): the on-device Frechet
        # distance can work from the rows (an n x n eigenproblem instead of two 2048 x 2048 ones)
        keep = int(args.num_samples) if (getattr(args, "fid_on_device", False) and args.num_samples < self.feature_dim) else 0
        acc = ActivationAccumulator(self.feature_dim, self._ev.device, keep_rows=keep) if self.features is not None else None
        if acc is not None and self._ref_dev is not None and self._ref_dev[0] is self.ref_stats:
            acc._ref_dev = self._ref_dev       # the reference statistics stay on the device across candidates
        host_images = []
        produced = 0
        batch_idx = 0
        # rounds still to run (this rank takes one batch per round), and how many of them ride in one pass over the networks:
        # images are bitwise those of separate passes (CandidateEvaluator.sample_batches), the chip is filled like the headline batch
        rounds = -(-args.num_samples // (args.batch_size * world))
        merge, per_pass = merge_policy(int(getattr(args, "image_size", 64)), args.batch_size, int(getattr(args, "merge_batches", 0) or 0), rounds)
        if merge > 1 and not getattr(self, "_merge_logged", False):   # once per search: log.txt shows the deviation from the reference's launch unit
            logger.log(f"evaluating {merge} batches of {args.batch_size} per pass over the networks ({per_pass} images per pass; "
                       "bitwise the images of separate passes; --merge_batches 1 restores the reference's launch unit)")
            self._merge_logged = True
        while produced < args.num_samples:
            k = min(merge, rounds - batch_idx)
            seeds = [seed0 + 7919 * ((batch_idx + j) * world + rank) for j in range(k)]
            u8s = [self._ev.sample_batch(args.batch_size, seed=seeds[0])] if k == 1 else self._ev.sample_batches(args.batch_size, seeds)
            for u8 in u8s:
                # the reference keeps arr[:num_samples] of the rank-major concatenation of every round
                start = produced + rank * args.batch_size
                keep = max(0, min(args.batch_size, args.num_samples - start))
                if acc is not None:
                    if keep > 0:
                        acc.add_from(self.features, u8[:keep])   # on a side stream, next to the next pass's sampling
                else:
                    host_images.append(u8[:keep].cpu().numpy())
                produced += args.batch_size * world
                batch_idx += 1
                logger.log('created ' + str(min(produced, batch_idx * args.batch_size * world)) + ' samples')
        if world > 1 or (not local and dist_util.collectives_on()):
            import torch.distributed as dist
            dist.barrier()
        logger.log("sampling complete")
        sample_time = time.time() - t1
        t1 = time.time()
        if acc is not None:
            if getattr(args, "fid_on_device", False):
                fid = acc.frechet_distance_device(self.ref_stats, None, local=local)  # eigh on the GPU instead of the host sqrtm
                self._ref_dev = acc._ref_dev
            else:
                fid = float(acc.statistics(None, local=local).frechet_distance(self.ref_stats))
        else:
            if world > 1:
                raise NotImplementedError("host-evaluator FID with several ranks: pass a device `features` function "
                                          "so that statistics are pooled on the GPUs")
            arr = np.concatenate(host_images, axis=0)[: args.num_samples]
            fid = float(cal_fid(arr, 64, self.evaluator, ref_stats=self.ref_stats))
        fid_time = time.time() - t1
        if acc is not None and acc.last_collective and not getattr(self, "_coll_logged", False):   # once per search
            c = acc.last_collective
            logger.log(f"collective: all_gather of the pooled FID statistics, backend {c['backend']}, {c['world_size']} rank(s), "
                       f"{c['bytes_per_rank']} B per rank, on {c['device']}")
            self._coll_logged = True
        logger.log('reset_time: ' + str(reset_time) + ', sample_time: ' + str(sample_time) + ', fid_time: ' + str(fid_time))
        self.last_times = {"reset_time": reset_time, "sample_time": sample_time, "fid_time": fid_time,
                           "images": int(args.num_samples), "batches_this_rank": batch_idx}
        return fid

    # ------------------------------------------------------------------ EA bookkeeping (reference order of RNG draws)
    def update_top_k(self, candidates, *, k, key, reverse=False):
        assert k in self.keep_top_k
        logger.log('select ......')
        t = self.keep_top_k[k]
        t += candidates
        t.sort(key=key, reverse=reverse)
        self.keep_top_k[k] = t[:k]

    def sample_active_subnet(self):
        if self.search_space is not None:
            use_timestep = self.search_space  # shuffled in place, as the reference does
        else:
            use_timestep = [i for i in range(self.base_diffusion.original_num_steps)]
        random.shuffle(use_timestep)
        return use_timestep[:self.time_step]

    def _visit(self, cand):
        if cand not in self.vis_dict:
            self.vis_dict[cand] = {}
        info = self.vis_dict[cand]
        if 'visited' in info:
            logger.log('cand: {} has visited!'.format(cand))
            return False
        if self.population_parallel:
            self._pending.append(cand)  # evaluated by flush_pending(), sharded over ranks
        else:
            info['fid'] = self.get_cand_fid(args=self.args, cand=eval(cand))
            logger.log('cand: {}, fid: {}'.format(cand, info['fid']) + self.fid_note)
        info['visited'] = True
        return True

    def candidate_cost(self, cand) -> int:
        """Relative cost of evaluating a candidate = UNet layer evaluations per image: every step costs `layer_num`
        minus its skipped layers (a plain timestep list: one unit per step)."""
        if isinstance(cand, dict):
            L = int(getattr(self.model, "layer_num", 0) or getattr(self.args, "layer_num", 0) or 1)
            return sum(max(1, L - len(sk)) for sk in cand["skip_layers"])
        return len(cand)

    @staticmethod
    def assign_candidates(costs, world):
        """Longest-processing-time-first: candidates sorted by cost (descending, ties by index) go one by one to the
        least-loaded rank (ties: lowest rank).  Deterministic, so every rank derives the same owner list with no
        communication; layer-skip candidates are cheaper, and round-robin left ranks finishing unevenly."""
        owner = [0] * len(costs)
        load = [0] * world
        for i in sorted(range(len(costs)), key=lambda j: (-costs[j], j)):
            r = min(range(world), key=lambda q: (load[q], q))
            owner[i] = r
            load[r] += costs[i]
        return owner

    def flush_pending(self):
        """Evaluate every queued candidate, each on ONE rank (cost-aware assignment); one all_gather of the FIDs."""
        if not self._pending:
            return
        import torch.distributed as dist
        multi = dist_util.collectives_on()   # > 1 rank (or a forced world-size-1 group: the same calls through RCCL on one GPU)
        world = dist.get_world_size() if multi else 1
        rank = dist.get_rank() if multi else 0
        pending, self._pending = self._pending, []
        cands = [eval(c) for c in pending]
        costs = [self.candidate_cost(c) for c in cands]
        owner = self.assign_candidates(costs, world)
        fids = np.zeros(len(pending), dtype=np.float64)
        t0 = time.time()
        for i, c in enumerate(cands):
            if owner[i] == rank:
                fids[i] = self.get_cand_fid(args=self.args, cand=c)
        mine_s = time.time() - t0
        coll = None
        if multi:
            dev = self._ev.device if (self._ev is not None and dist.get_backend() == "nccl") else torch.device("cpu")
            mine = torch.from_numpy(fids).to(dev)
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            coll = {"op": "all_gather", "backend": dist.get_backend(), "world_size": world, "bytes_per_rank": int(mine.numel() * 8),
                    "device": str(mine.device)}
            for i in range(len(pending)):
                fids[i] = float(parts[owner[i]][i])
            logger.log(f"collective: all_gather of {len(pending)} candidate FIDs, backend {coll['backend']}, {world} rank(s), on {coll['device']}")
        # what this epoch's evaluation looked like from this rank (bench.py --workload population reports it per rank)
        self.last_flush = {"candidates": len(pending), "owner": owner, "costs": costs,
                           "assigned": sum(1 for o in owner if o == rank), "assigned_cost": sum(c for c, o in zip(costs, owner) if o == rank),
                           "evaluate_s": mine_s, "collective": coll}
        for cand, fid in zip(pending, fids):
            self.vis_dict[cand]['fid'] = float(fid)
            logger.log('cand: {}, fid: {}'.format(cand, float(fid)) + self.fid_note)

    def is_legal_before_search(self, cand):
        return self._visit(cand)

    def is_legal(self, cand):
        return self._visit(cand)

    def _fill_random(self, num, legal):
        logger.log('random select ........')
        while len(self.candidates) < num:
            cand = str(self.sample_active_subnet())
            if not legal(cand):
                continue
            self.candidates.append(cand)
            logger.log('random {}/{}'.format(len(self.candidates), num))
        logger.log('random_num = {}'.format(len(self.candidates)))

    def get_random_before_search(self, num):
        self._fill_random(num, self.is_legal_before_search)

    def get_random(self, num):
        self._fill_random(num, self.is_legal)

    def get_cross(self, k, cross_num):
        assert k in self.keep_top_k
        logger.log('cross ......')
        res = []
        max_iters = cross_num * 10
        while len(res) < cross_num and max_iters > 0:
            max_iters -= 1
            cand1 = eval(choice(self.keep_top_k[k]))
            cand2 = eval(choice(self.keep_top_k[k]))
            new_cand = [cand1[i] if np.random.random_sample() < 0.5 else cand2[i] for i in range(len(cand1))]
            cand = str(new_cand)
            if not self.is_legal(cand):
                continue
            res.append(cand)
            logger.log('cross {}/{}'.format(len(res), cross_num))
        logger.log('cross_num = {}'.format(len(res)))
        return res

    def _mutate(self, cand, m_prob):
        if self.search_space is not None:
            all_index = self.search_space
        else:
            all_index = range(self.base_diffusion.original_num_steps)
        candidates = [i for i in all_index if i not in cand]
        for i in range(len(cand)):
            if np.random.random_sample() < m_prob:
                new_c = random.choice(candidates)
                del candidates[candidates.index(new_c)]
                cand[i] = new_c
                if len(candidates) == 0:
                    break
        return cand

    def get_mutation(self, k, mutation_num, m_prob):
        assert k in self.keep_top_k
        logger.log('mutation ......')
        res = []
        max_iters = mutation_num * 10
        while len(res) < mutation_num and max_iters > 0:
            max_iters -= 1
            cand = str(self._mutate(eval(choice(self.keep_top_k[k])), m_prob))
            if not self.is_legal(cand):
                continue
            res.append(cand)
            logger.log('mutation {}/{}'.format(len(res), mutation_num))
        logger.log('mutation_num = {}'.format(len(res)))
        return res

    def mutate_init_x(self, x0, mutation_num, m_prob):
        logger.log('mutation x0 ......')
        res = []
        max_iters = mutation_num * 10
        while len(res) < mutation_num and max_iters > 0:
            max_iters -= 1
            cand = str(self._mutate(eval(x0), m_prob))
            if not self.is_legal_before_search(cand):
                continue
            res.append(cand)
            logger.log('mutation x0 {}/{}'.format(len(res), mutation_num))
        logger.log('mutation_num = {}'.format(len(res)))
        return res

    def search(self):
        args = self.args
        logger.log('population_num = {} select_num = {} mutation_num = {} crossover_num = {} random_num = {} max_epochs = {}'.format(
            self.population_num, self.select_num, self.mutation_num, self.crossover_num,
            self.population_num - self.mutation_num - self.crossover_num, self.max_epochs))
        if self.x0 != '':  # search_uncondition_model.py:509-512
            self.get_random_before_search(self.population_num // 2)
            self.candidates += self.mutate_init_x(x0=self.x0, mutation_num=self.population_num - self.population_num // 2,
                                                  m_prob=0.05)
        elif getattr(args, "use_ddim_init_x", False) is False:
            self.get_random_before_search(self.population_num)
        else:
            steps = self.base_diffusion.original_num_steps
            timestep_respacing = ('ddim' if args.use_ddim else '') + str(args.time_step)
            init_x = str(list(space_timesteps(steps, timestep_respacing)))
            self.is_legal_before_search(init_x)
            self.candidates.append(init_x)
            # the guided script seeds pop//2 + 1 random candidates, the unconditional one pop//2
            extra = 0 if self.break_at_last_epoch else 1
            self.get_random_before_search(self.population_num // 2 + extra)
            self.candidates += self.mutate_init_x(x0=init_x, mutation_num=self.population_num - self.population_num // 2 - 1,
                                                  m_prob=0.1)
        while self.epoch < self.max_epochs:
            logger.log('epoch = {}'.format(self.epoch))
            self.flush_pending()
            self.update_top_k(self.candidates, k=self.select_num, key=lambda x: self.vis_dict[x]['fid'])
            self.update_top_k(self.candidates, k=50, key=lambda x: self.vis_dict[x]['fid'])
            logger.log('epoch = {} : top {} result'.format(self.epoch, len(self.keep_top_k[50])))
            for i, cand in enumerate(self.keep_top_k[50]):
                logger.log('No.{} {} fid = {}'.format(i + 1, cand, self.vis_dict[cand]['fid']) + self.fid_note)
            if self.break_at_last_epoch and self.epoch + 1 == self.max_epochs:
                break
            self.candidates = self.get_mutation(self.select_num, self.mutation_num, self.m_prob)
            self.candidates += self.get_cross(self.select_num, self.crossover_num)
            self.get_random(self.population_num)
            self.epoch += 1
        self.flush_pending()


class DynamicEvolutionSearcher(EvolutionSearcher):
    """Joint search over timesteps AND per-step layer-skip lists on the dynamic UNet: the reference's
    search_dynamic_unet_imagenet64_classifier_guidance_progressive.py (EvolutionSearcher, :155-715).

    A candidate is ``{'timesteps': [t_0, ...], 'skip_layers': [[layer ids skipped at t_0], ...]}``; its budget is
    ``max_index_number`` = time_step * layer_num evaluated layers (or ``index_step``).  The fraction of layers a step may
    skip is drawn from ``skip_layer_range``, which the search opens progressively: [0, 0] until the best candidate
    stalls (or epoch 5), then the upper end grows by max_prun / 5 per epoch up to max_prun, and from epoch 6 the lower end
    is min_prun (:684-692).  Every operator consumes ``random`` / ``np.random`` in the reference's order, including its
    quirks -- the list comparison that decides whether a crossover child is padded from a parent (:494-500) and the
    skip-list mutation that draws its replacement but leaves the list unchanged (``==`` at :568, :632) -- because the
    candidate sequence is the data contract (pinned by tests/golden/ea_dynamic_trajectory.npz).
    """

    def __init__(self, args, model, base_diffusion, time_step, classifier=None, index_step=None, **kw):
        kw.setdefault("variant", "guided")
        super().__init__(args, model, base_diffusion, time_step, classifier=classifier, **kw)
        self.init_time_step = time_step
        self.model_layers = model.layer_num if model is not None else int(getattr(args, "layer_num"))
        self.max_index_number = time_step * self.model_layers
        if index_step is not None:
            self.max_index_number = eval(index_step) if isinstance(index_step, str) else int(index_step)
        self.max_prun = args.max_prun
        self.min_prun = args.min_prun
        self.skip_layer_range = [0, 0]
        self.last_best_cand = None

    def cand2gen(self, cand):
        """Flat index encoding (layer + layer_num * timestep of every evaluated layer), zero-padded (:207-217)."""
        ret = []
        for t, skipped in zip(cand['timesteps'], cand['skip_layers']):
            kept = [k for k in range(self.model_layers) if k not in skipped]
            ret += [k + self.model_layers * t for k in kept]
        return ret + [0] * max(0, self.max_index_number - len(ret))

    # ------------------------------------------------------------------ operators (reference order of RNG draws)
    def _draw_skips(self, count):
        layers = [i for i in range(self.model_layers)]
        random.shuffle(layers)
        return layers[:count]

    def sample_active_subnet(self):
        lo, hi = self.skip_layer_range
        L, budget = self.model_layers, self.max_index_number
        use_timestep = [i for i in range(self.base_diffusion.original_num_steps)]
        random.shuffle(use_timestep)
        used, timesteps, skips = 0, [], []
        while True:
            n_skip = None
            tries = 0
            while n_skip is None or used + L - n_skip > budget:   # redraw until this step fits the budget
                n_skip = int((np.random.random_sample() * (hi - lo) + lo) * L)
                tries += 1
                if tries > 10 ** 6:
                    raise RuntimeError("sample_active_subnet: no skip count fits the index budget")
            skips.append(self._draw_skips(n_skip))
            timesteps.append(use_timestep[len(timesteps)])
            used += L - n_skip
            least = L - int(L * hi)                                # the cheapest step still allowed
            if used + least > budget:
                break
            if used + least == budget:                             # exactly one maximally pruned step fits
                skips.append(self._draw_skips(int(L * hi)))
                timesteps.append(use_timestep[len(timesteps)])
                break
        return {'timesteps': timesteps, 'skip_layers': skips}

    def _mutate_timesteps(self, cand, m_prob):
        pool = [i for i in range(self.base_diffusion.original_num_steps) if i not in cand['timesteps']]
        for i in range(len(cand['timesteps'])):
            if np.random.random_sample() < m_prob:
                new_c = random.choice(pool)
                del pool[pool.index(new_c)]
                cand['timesteps'][i] = new_c
                if len(pool) == 0:
                    break

    def _touch_skips(self, skipped, m_prob):
        """The reference draws a replacement layer per mutated entry but never stores it: RNG draws only."""
        pool = [j for j in range(self.model_layers) if j not in skipped]
        for _ in range(len(skipped)):
            if np.random.random_sample() < m_prob:
                new_c = random.choice(pool)
                del pool[pool.index(new_c)]
                if len(pool) == 0:
                    break

    def _mutate(self, cand, m_prob, fill_empty=True):
        self._mutate_timesteps(cand, m_prob)
        if self.skip_layer_range[1] == 0:
            return cand
        lo, hi = self.skip_layer_range
        for i in range(len(cand['skip_layers'])):
            if fill_empty and len(cand['skip_layers'][i]) == 0:
                if np.random.random_sample() < m_prob:             # an unpruned step gets a fresh skip list
                    n_skip = int((np.random.random_sample() * (hi - lo) + lo) * self.model_layers)
                    cand['skip_layers'][i] = self._draw_skips(n_skip)
            else:
                self._touch_skips(cand['skip_layers'][i], m_prob)
        return cand

    def get_cross(self, k, cross_num):
        assert k in self.keep_top_k
        logger.log('cross ......')
        res = []
        max_iters = cross_num * 10
        while len(res) < cross_num and max_iters > 0:
            max_iters -= 1
            cand1 = eval(choice(self.keep_top_k[k]))
            cand2 = eval(choice(self.keep_top_k[k]))
            child = {'timesteps': [], 'skip_layers': []}
            for i in range(min(len(cand1['timesteps']), len(cand2['timesteps']))):
                src = cand1 if np.random.random_sample() < 0.5 else cand2
                child['timesteps'].append(src['timesteps'][i])
                child['skip_layers'].append(src['skip_layers'][i])
            for parent in (cand1, cand2):                          # list (lexicographic) comparison, as in the reference
                if child['timesteps'] < parent['timesteps']:
                    child['timesteps'] += parent['timesteps'][len(child['timesteps']):]
                    child['skip_layers'] += parent['skip_layers'][len(child['skip_layers']):]
            cand = str(child)
            if not self.is_legal(cand):
                continue
            res.append(cand)
            logger.log('cross {}/{}'.format(len(res), cross_num))
        logger.log('cross_num = {}'.format(len(res)))
        return res

    def mutate_init_x(self, x0, mutation_num, m_prob):
        logger.log('mutation x0 ......')
        res = []
        max_iters = mutation_num * 10
        while len(res) < mutation_num and max_iters > 0:
            max_iters -= 1
            cand = str(self._mutate(eval(x0), m_prob, fill_empty=False))
            if not self.is_legal_before_search(cand):
                continue
            res.append(cand)
            logger.log('mutation x0 {}/{}'.format(len(res), mutation_num))
        logger.log('mutation_num = {}'.format(len(res)))
        return res

    def search(self):
        args = self.args
        logger.log('population_num = {} select_num = {} mutation_num = {} crossover_num = {} random_num = {} max_epochs = {}'.format(
            self.population_num, self.select_num, self.mutation_num, self.crossover_num,
            self.population_num - self.mutation_num - self.crossover_num, self.max_epochs))
        if getattr(args, "use_ddim_init_x", False) is False:
            self.get_random_before_search(self.population_num)
        else:
            steps = self.base_diffusion.original_num_steps
            init_x = list(space_timesteps(steps, ('ddim' if args.use_ddim else '') + str(args.time_step)))
            init_cand = str({'timesteps': init_x, 'skip_layers': [[] for _ in init_x]})
            self.is_legal_before_search(init_cand)
            self.candidates.append(init_cand)
            self.get_random_before_search(self.population_num // 2 + 1)
            self.candidates += self.mutate_init_x(x0=init_cand, mutation_num=self.population_num - self.population_num // 2 - 1,
                                                  m_prob=0.1)
        while self.epoch < self.max_epochs:
            logger.log('epoch = {}'.format(self.epoch))
            self.flush_pending()
            self.update_top_k(self.candidates, k=self.select_num, key=lambda x: self.vis_dict[x]['fid'])
            self.update_top_k(self.candidates, k=50, key=lambda x: self.vis_dict[x]['fid'])
            logger.log('epoch = {} : top {} result'.format(self.epoch, len(self.keep_top_k[50])))
            for i, cand in enumerate(self.keep_top_k[50]):
                logger.log('No.{} {} fid = {}'.format(i + 1, cand, self.vis_dict[cand]['fid']) + self.fid_note)
            best = self.keep_top_k[50][0]
            if self.skip_layer_range[1] == 0 and (self.last_best_cand == best or self.epoch > 4):
                self.skip_layer_range[1] = self.max_prun / 5
            elif 0 < self.skip_layer_range[1] < self.max_prun:
                self.skip_layer_range[1] += self.max_prun / 5
            if self.skip_layer_range[0] == 0 and self.epoch > 5:
                self.skip_layer_range[0] = self.min_prun
            self.last_best_cand = best
            logger.log('skip_layer_range_left = {} , skip_layer_range_right {}'.format(*self.skip_layer_range))
            if self.epoch + 1 == self.max_epochs:
                break
            self.candidates = self.get_mutation(self.select_num, self.mutation_num, self.m_prob)
            self.candidates += self.get_cross(self.select_num, self.crossover_num)
            self.get_random(self.population_num)
            self.epoch += 1
        self.flush_pending()
