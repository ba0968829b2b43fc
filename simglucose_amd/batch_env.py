"""BatchedT1DSimEnv: N independent T1DSimEnv episodes advanced by one HIP kernel launch per step.

The host side holds the batched state as struct-of-arrays PyTorch-ROCm tensors (env index fastest)
and hands their device pointers to libt1d_hip.so through the C ABI of include/t1d.h.  Semantics per
env are those of the reference's ``simglucose/simulation/env.py`` ``T1DSimEnv.reset/step`` composed
of ``T1DPatient`` + ``CGMSensor`` + ``InsulinPump`` + a meal scenario; instead of SciPy's adaptive DOPRI5 the ODE is
integrated by the library's split scheme built on ``n_sub`` sub-steps per minute, with per-minute step sizes chosen by a
deterministic rule (``adaptive_gut``), or by classical RK4 (``set_option("integrator", 0)``).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, params

_STATE_KEYS = ("x", "planned", "last_qsto", "last_food", "t", "meta", "episode", "next_meal", "last_cgm", "ar_e",
               "pts", "prev_risk", "dbar")
_OUT_KEYS = ("cgm", "bg", "reward", "done", "lbgi", "hbgi", "risk", "meal", "insulin", "cgm0")


class BatchedT1DSimEnv:
    """A batch of ``n`` glucose-insulin simulation environments on one MI355X.

    patient: one name, a list of names (len n) or an integer array of table rows (len n).
    sensor / pump: hardware names from the parameter tables (one per batch).
    noise: "philox" draws the CGM noise normals in-kernel (rocRAND Philox4x32-10, stream =
           global env id); "host" reads them from ``normals[n_draws, n]`` (exact parity with
           ``numpy.random.RandomState(seed).randn()`` streams supplied by the caller).
    cgm_history: keep the last hour of observations per env on the device -- the ``CGM_hist[-window:]`` that
           ``T1DSimEnv.step`` hands to a custom ``reward_fun`` (simulation/env.py:100-102), window = 60 / sample_time
           samples -- so that ``step(..., reward_fun=f)`` works on the batch: ``f(window)`` gets a ``[window, n]``
           tensor, oldest sample first, NaN where an episode is younger than that, and returns ``[n]`` rewards.
           Off by default: the fused default reward (risk_diff) needs only the previous sample.
    """

    def __init__(self, patient="adolescent#001", n_envs=None, sensor="Dexcom", pump="Insulet",
                 dtype=torch.float64, device="cuda:0", n_sub=4, seed=0, env_offset=0, noise="philox",
                 normals=None, random_init_bg=False, extra_outputs=True, sensor_row=None, pump_row=None,
                 patient_table=None, use_pump=True, adaptive_gut=True, cgm_history=False):
        self._L = _lib.lib()                     # raises T1DError if the HIP extension is missing
        if not torch.cuda.is_available():
            raise _lib.T1DError("BatchedT1DSimEnv needs a ROCm GPU (torch.cuda.is_available() is False)")
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("dtype must be torch.float64 or torch.float32")
        self.device = torch.device(device)
        self.dtype = dtype
        self.n_sub = int(n_sub)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.env_offset = int(env_offset)
        self.random_init_bg = bool(random_init_bg)
        self.names, self.table = params.patient_table()
        if patient_table is not None:            # caller-supplied rows (T1D_P_* column order), e.g. edited parameters
            self.table = np.ascontiguousarray(np.atleast_2d(np.asarray(patient_table, dtype=np.float64)))
            if self.table.shape[1] != _lib.P_NCOLS:
                raise ValueError("patient_table must have %d columns" % _lib.P_NCOLS)
            self.names = ["custom#%03d" % k for k in range(self.table.shape[0])]
        if isinstance(patient, str) and patient_table is not None:
            patient = np.zeros(int(n_envs or 1), dtype=np.int64)
        if isinstance(patient, str):
            if n_envs is None:
                n_envs = 1
            pid = np.full(int(n_envs), params.patient_index(patient), dtype=np.int64)
        else:
            arr = list(patient) if not isinstance(patient, (np.ndarray, torch.Tensor)) else patient
            if len(arr) and isinstance(arr[0], str):
                pid = np.array([params.patient_index(p) for p in arr], dtype=np.int64)
            else:
                pid = np.asarray(torch.as_tensor(arr).cpu().numpy(), dtype=np.int64)
            if n_envs is not None and int(n_envs) != len(pid):
                raise ValueError("n_envs does not match the patient list")
        if pid.size < 1 or pid.min() < 0 or pid.max() >= len(self.names):
            raise ValueError("patient index out of range")
        self.n = int(pid.size)
        self.patient_idx = pid
        self.sensor_name, self.pump_name = sensor, pump
        self.sensor_row = np.asarray(sensor_row if sensor_row is not None else params.sensor_row(sensor), dtype=np.float64)
        self.pump_row = np.asarray(pump_row if pump_row is not None else params.pump_row(pump), dtype=np.float64)
        self.sample_time = float(self.sensor_row[5])
        self.minutes_per_step = int(self.sample_time)

        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._ctx = C.c_void_p()
        dp = C.POINTER(C.c_double)
        tab = np.ascontiguousarray(self.table)
        _lib.check(self._L.t1d_ctx_create(dev_index, tab.ctypes.data_as(dp), tab.shape[0], tab.shape[1],
                                          self.sensor_row.ctypes.data_as(dp), self.pump_row.ctypes.data_as(dp),
                                          C.byref(self._ctx)))
        # split integrator: half-size gut steps in the minutes that cross a gastric-emptying transition fast (library default)
        self.set_option("adaptive_gut", 1 if adaptive_gut else 0)
        n, dv, ft = self.n, self.device, dtype
        z = lambda *shape, dt=ft: torch.zeros(*shape, dtype=dt, device=dv)
        # packed state (include/t1d.h): one [45, n] float buffer and one [4, n] int32 buffer; the named
        # tensors are views, so every staged row is one base pointer plus a 32-bit offset on the device
        self.state = z(45, n)
        self.x = self.state[0:13]; self.planned = self.state[13]; self.last_qsto = self.state[14]
        self.last_food = self.state[15]; self.last_cgm = self.state[16]; self.prev_risk = self.state[17]
        self.pts = self.state[18:44]; self.dbar = self.state[44]
        self.istate = z(4, n, dt=torch.int32)
        self.t = self.istate[0]; self.meta = self.istate[1]; self.next_meal = self.istate[2]; self.episode = self.istate[3]
        self.meta.copy_(torch.from_numpy(pid.astype(np.int32)))            # patient row in bits 0-7
        self.next_meal.fill_(_lib.MEAL_UNUSED)
        self.ar_e = z(n)
        self.cgm = z(n); self.bg = z(n); self.reward = z(n); self.done = z(n, dt=torch.uint8)
        self.cgm0 = z(n)                   # CGM sample #0 of the current episode (CGM_hist[0]); written by reset only
        if extra_outputs:
            self.lbgi = z(n); self.hbgi = z(n); self.risk = z(n); self.meal = z(n); self.insulin = z(n)
        else:
            self.lbgi = self.hbgi = self.risk = self.meal = self.insulin = None
        self._zero_action = z(n)
        self._basal_buf = z(n); self._bolus_buf = z(n)
        self.meal_time = None; self.meal_amt = None
        self.normals = None
        self._b = _lib.Batch()
        b = self._b
        b.n, b.env_offset, b.dtype, b.seed = n, self.env_offset, (_lib.T1D_F64 if ft == torch.float64 else _lib.T1D_F32), self.seed
        for k in _STATE_KEYS + _OUT_KEYS:
            tns = getattr(self, k)
            setattr(b, k, tns.data_ptr() if tns is not None else None)
        b.n_meals = 0; b.n_normals = 0
        b.flags = 0 if use_pump else _lib.T1D_BATCH_NO_PUMP
        if noise not in ("philox", "host"):
            raise ValueError("noise must be 'philox' or 'host'")
        self.noise = noise
        if noise == "host":
            if normals is None:
                raise ValueError("noise='host' needs normals[n_draws, n]")
            self.set_normals(normals)
        self._closed = False
        self._clock = None         # minutes since the last FULL reset while every env shares one clock, else None
        self._iver = None          # version counter of the integer state tensors when the shadow clock was last valid
        self._flags0 = b.flags
        self.window = int(60 / self.sample_time)            # samples a custom reward function sees (env.py:100)
        self._hist = self._hist_pos = self._hist_cnt = None
        if cgm_history:
            self.enable_cgm_history()

    # ------------------------------------------------------------------ inputs
    def _as_input(self, v, buf):
        """-> a contiguous [n] tensor of the env dtype on the env device (copying only if needed)."""
        if isinstance(v, torch.Tensor) and v.dtype == self.dtype and v.device == self.device and v.shape == (self.n,) \
                and v.is_contiguous():
            return v
        buf.copy_(torch.as_tensor(v, dtype=self.dtype, device=self.device).expand(self.n))
        return buf

    def set_normals(self, normals):
        """Host-supplied standard normals [n_draws, n]: row 0 seeds the AR(1) state, rows 1.. are
        consumed ten per 150-minute noise block (noise_gen.py:84-97)."""
        t = torch.as_tensor(normals, dtype=self.dtype).to(self.device).contiguous()
        if t.dim() != 2 or t.shape[1] != self.n:
            raise ValueError("normals must have shape [n_draws, n]")
        self.normals = t
        self._b.normals = t.data_ptr(); self._b.n_normals = t.shape[0]
        self.noise = "host"

    def set_meals(self, meal_time, meal_amt):
        """Per-env meal table: meal_time[m, i] = minute since episode start (ascending per env,
        unused = MEAL_UNUSED), meal_amt[m, i] = grams announced at that minute."""
        mt = torch.as_tensor(meal_time).to(torch.int32).to(self.device).contiguous()
        ma = torch.as_tensor(meal_amt).to(self.dtype).to(self.device).contiguous()
        if mt.dim() != 2 or mt.shape != ma.shape or mt.shape[1] != self.n:
            raise ValueError("meal tables must both have shape [n_meals, n]")
        self.meal_time, self.meal_amt = mt, ma
        self._b.meal_time, self._b.meal_amt, self._b.n_meals = mt.data_ptr(), ma.data_ptr(), mt.shape[0]
        # restart the table scan: cursor 0, next entry = row 0 (rows before the current minute are skipped lazily)
        self.meta.bitwise_and_(0xFFFF)
        self.next_meal.copy_(mt[0])

    def set_option(self, name, value):
        """t1d_ctx_set_option (include/t1d.h): e.g. ("integrator", 0) = classical RK4, ("adaptive_gut", 0) = the split
        scheme at level 1 in every minute, ("math", 0) = the ocml-tanh / IEEE-division RHS with classical RK4."""
        _lib.check(self._L.t1d_ctx_set_option(self._ctx, name.encode(), int(value)))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ CGM history for custom reward functions
    def enable_cgm_history(self):
        """allocate the per-env ring of the last `window` observations (call before reset())"""
        if self._hist is None:
            self._hist = torch.full((self.window, self.n), float("nan"), dtype=self.dtype, device=self.device)
            self._hist_pos = torch.zeros(self.n, dtype=torch.int64, device=self.device)
            self._hist_cnt = torch.zeros(self.n, dtype=torch.int64, device=self.device)

    def _hist_reset(self, mask):
        """T1DSimEnv._reset: CGM_hist = [sample #0] (env.py:126), which the reset kernel leaves in cgm0"""
        if self._hist is None:
            return
        m = torch.ones(self.n, dtype=torch.bool, device=self.device) if mask is None else mask.bool()
        self._hist[:, m] = float("nan")
        self._hist_pos[m] = 0
        self._hist_cnt[m] = 1
        self._hist[0, m] = self.cgm0[m]

    def _hist_push(self):
        """CGM_hist.append(CGM) (env.py:94) for every env"""
        self._hist_pos = (self._hist_pos + 1) % self.window
        self._hist.scatter_(0, self._hist_pos.unsqueeze(0), self.cgm.unsqueeze(0))
        self._hist_cnt = torch.clamp(self._hist_cnt + 1, max=self.window)

    def _hist_after_rollout(self):
        """a roll-out advances many steps inside one launch: what the ring holds afterwards is the last observation only"""
        if self._hist is not None:
            self._hist.fill_(float("nan"))
            self._hist_pos.zero_()
            self._hist_cnt.fill_(1)
            self._hist[0] = self.cgm

    def cgm_window(self):
        """-> [window, n]: CGM_hist[-window:] of every env, oldest first, NaN-padded at the top while an episode has fewer
        samples (the reference passes a shorter list then)."""
        if self._hist is None:
            raise _lib.T1DError("the CGM history is off: construct with cgm_history=True (or call enable_cgm_history() before reset())")
        k = torch.arange(self.window, device=self.device).unsqueeze(1)
        w = torch.gather(self._hist, 0, (self._hist_pos.unsqueeze(0) + 1 + k) % self.window)
        return torch.where(k >= self.window - self._hist_cnt.unsqueeze(0), w, torch.full_like(w, float("nan")))

    # ------------------------------------------------------------------ reset / step
    def reset(self, mask=None, x0=None):
        """T1DSimEnv.reset() on every env (or those with mask != 0).  -> observation CGM [n]."""
        b = self._b
        keep = None
        if x0 is not None:
            keep = torch.as_tensor(x0, dtype=self.dtype).to(self.device).contiguous()
            if keep.shape != (13, self.n):
                raise ValueError("x0 must have shape [13, n]")
            b.x0_override = keep.data_ptr()
        else:
            b.x0_override = None
        mptr = None
        if mask is not None:
            mask = torch.as_tensor(mask).to(self.device).to(torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        self._clock = 0 if mask is None else None
        with torch.cuda.device(self.device):
            _lib.check(self._L.t1d_reset(self._ctx, C.byref(b), mptr, int(self.random_init_bg), self._stream()))
        self._iver = self.istate._version
        b.x0_override = None
        self._keep = (keep, mask)
        self._hist_reset(mask)
        return self.cgm

    def step(self, basal, bolus=None, cho=None, minutes=None, reward_fun=None):
        """One env.step for the whole batch: a single kernel launch advancing ``minutes``
        (default int(sample_time)) with the action held.  -> (obs CGM [n], reward [n], done [n], info).
        reward_fun: None = the fused default, risk_diff (env.py:27-33); a callable gets ``cgm_window()`` -- the batch
        form of ``reward_fun(CGM_hist[-window:])``, env.py:100-102 -- and returns the rewards [n] (needs cgm_history)."""
        b = self._b
        minutes = self.minutes_per_step if minutes is None else int(minutes)
        bas = self._as_input(basal, self._basal_buf)
        b.basal = bas.data_ptr()
        if bolus is None:
            b.bolus = None
        else:
            b.bolus = self._as_input(bolus, self._bolus_buf).data_ptr()
        if cho is not None:
            cho = torch.as_tensor(cho, dtype=self.dtype).to(self.device).contiguous()
            if cho.shape != (minutes, self.n):
                raise ValueError("cho must have shape [minutes, n]")
            b.cho = cho.data_ptr()
        else:
            b.cho = None
        b.flags = self._flags0
        if self._clock is not None and self.istate._version != self._iver:
            # somebody wrote env.t / env.meta / env.next_meal through torch since the shadow clock was taken (kernel
            # launches do not move the version counter): the envs' own clocks decide again
            self._clock = None
        if self._clock is not None:
            # every env shares the clock: the host knows whether a sample in (t, t + minutes] opens a noise block
            st, S = self.minutes_per_step, int(150 // self.sample_time)
            due = any((t1 % st == 0) and ((1 + t1 // st) % S == 0) for t1 in range(self._clock + 1, self._clock + minutes + 1))
            if not due:
                b.flags = self._flags0 | _lib.T1D_BATCH_NO_REFILL_DUE
        clock, self._clock = self._clock, None        # the shadow clock survives only a call that went through
        with torch.cuda.device(self.device):
            _lib.check(self._L.t1d_step(self._ctx, C.byref(b), minutes, self.n_sub, self._stream()))
        if clock is not None:
            self._clock = clock + minutes
        self._keep = (bas, cho)
        if self._hist is not None:
            self._hist_push()
        if reward_fun is not None:
            reward = torch.as_tensor(reward_fun(self.cgm_window()), dtype=self.dtype, device=self.device).expand(self.n)
            return self.cgm, reward, self.done, self.info()
        return self.cgm, self.reward, self.done, self.info()

    def info(self):
        """live views of the device outputs and state: read-only for the caller (the reference's info['patient_state']
        is the solver's own array too, env.py:112).  The wrapper shadows the clock on the host to skip the noise-block
        refill pre-kernel; an in-place torch edit of env.t / env.meta / env.next_meal is noticed (the tensors' version
        counter) and drops the shadow, so that such an edit cannot leave a stale noise block behind -- for writes the
        counter cannot see there is invalidate_clock().  (A copy of t per step would cost a 4 MB device copy per launch at
        1 Mi envs: 6 % of the step.)"""
        return {"sample_time": self.sample_time, "bg": self.bg, "lbgi": self.lbgi, "hbgi": self.hbgi,
                "risk": self.risk, "meal": self.meal, "insulin": self.insulin, "patient_state": self.x,
                "t": self.t}

    def invalidate_clock(self):
        """forget the host's shadow clock: the next steps check every env's own clock for due noise-block refills again.
        (In-place torch edits of env.t / env.meta / env.next_meal are noticed through the tensors' version counter; this
        is for writes the counter cannot see, e.g. through a raw pointer.)"""
        self._clock = None

    @staticmethod
    def _set_trace(p, trace, n_steps):
        """trace: None or dict with any of bg, cgm, cho, insulin (tensors [rows, n]) and row = first row to write
        (see new_trace)"""
        p.bg_trace = p.cgm_trace = p.cho_trace = p.insulin_trace = None
        p.trace_row = 0
        if trace:
            row = int(trace.get("row", 0))
            for k, f in (("bg", "bg_trace"), ("cgm", "cgm_trace"), ("cho", "cho_trace"), ("insulin", "insulin_trace")):
                if trace.get(k) is not None:
                    if trace[k].shape[0] < row + n_steps or not trace[k].is_contiguous():
                        raise ValueError("trace['%s'] needs at least row + n_steps contiguous rows" % k)
                    setattr(p, f, trace[k].data_ptr())
            p.trace_row = row
            trace["row"] = row + int(n_steps)

    def new_trace(self, n_steps, columns=("bg", "cgm", "cho", "insulin")):
        """Device-resident history for the next `n_steps` roll-out steps, laid out as T1DSimEnv's history lists
        (simulation/env.py:119-155,169-180): row 0 holds what reset() recorded (BG0 and CGM sample #0; CHO and
        insulin have no row for the last time stamp, so their row r is the action of step r), row r >= 1 step r.
        Call right after reset(); pass the dict as rollout_*(trace=...)."""
        tr = {"row": 1}
        for k in columns:
            tr[k] = torch.full((int(n_steps) + 1, self.n), float("nan"), dtype=self.dtype, device=self.device)
        if "bg" in tr:
            tr["bg"][0] = self.bg
        if "cgm" in tr:
            tr["cgm"][0] = self.cgm0
        return tr

    def rollout_pid(self, n_steps, P, I, D, target=140.0, pid_state=None, stats=None, trace=None):
        """n_steps closed-loop PID steps in one launch (PIDController.policy + env.step per step).
        pid_state: dict(integ, prev) tensors [n] (created zeroed if None).  stats: optional dict
        with any of sum_risk, min_bg, max_bg (float [n]) and n_low, n_high (int32 [n])."""
        if pid_state is None:
            pid_state = {"integ": torch.zeros(self.n, dtype=self.dtype, device=self.device),
                         "prev": torch.zeros(self.n, dtype=self.dtype, device=self.device)}
        p = _lib.Pid()
        p.P, p.I, p.D, p.target = float(P), float(I), float(D), float(target)
        p.integ, p.prev = pid_state["integ"].data_ptr(), pid_state["prev"].data_ptr()
        stats = stats or {}
        for k in ("sum_risk", "min_bg", "max_bg", "n_low", "n_high"):
            setattr(p, k, stats[k].data_ptr() if k in stats else None)
        self._set_trace(p, trace, n_steps)
        self._b.cho = None
        self._b.flags = self._flags0
        clock, self._clock = self._clock, None
        with torch.cuda.device(self.device):
            _lib.check(self._L.t1d_rollout_pid(self._ctx, C.byref(self._b), C.byref(p), int(n_steps),
                                               self.minutes_per_step, self.n_sub, self._stream()))
        if clock is not None:
            self._clock = clock + int(n_steps) * self.minutes_per_step
        self._hist_after_rollout()
        return pid_state

    def bb_constants(self):
        """Per-env BBController constants (basal_bolus_ctrller.py:54-64): basal = u2ss*BW/6000 U/min, CR and CF from
        Quest.csv; patients Quest.csv does not list get the reference's 'Average' row (CR 1/15, CF 1/50,
        u2ss 1.43, BW 57).  -> dict of tensors [n]."""
        quest = params.quest_table()
        basal = np.empty(len(self.names)); cr = np.empty(len(self.names)); cf = np.empty(len(self.names))
        for k, name in enumerate(self.names):
            if name in quest:
                cr[k], cf[k] = quest[name][0], quest[name][1]
                basal[k] = params.basal_rate(self.table[k])
            else:
                cr[k], cf[k], basal[k] = 1.0 / 15.0, 1.0 / 50.0, 1.43 * 57.0 / 6000.0
        mk = lambda v: torch.as_tensor(v[self.patient_idx], dtype=self.dtype, device=self.device).contiguous()
        return {"basal": mk(basal), "cr": mk(cr), "cf": mk(cf)}

    def rollout_bb(self, n_steps, target=140.0, bb_state=None, stats=None, trace=None):
        """n_steps closed-loop BBController steps in one launch (SimObj.simulate with BBController: policy from
        the previous observation and the previous step's announced meal, then env.step).  bb_state: dict with
        basal, cr, cf (see bb_constants) and prev_meal [n] (created if None; prev_meal = 0 right after reset).
        Meals come from the meal tables (set_meals).  stats as in rollout_pid."""
        if bb_state is None:
            bb_state = self.bb_constants()
            bb_state["prev_meal"] = torch.zeros(self.n, dtype=self.dtype, device=self.device)
        p = _lib.Bb()
        p.target = float(target)
        for k in ("basal", "cr", "cf", "prev_meal"):
            setattr(p, k, bb_state[k].data_ptr())
        stats = stats or {}
        for k in ("sum_risk", "min_bg", "max_bg", "n_low", "n_high"):
            setattr(p, k, stats[k].data_ptr() if k in stats else None)
        self._set_trace(p, trace, n_steps)
        self._b.cho = None
        self._b.flags = self._flags0
        clock, self._clock = self._clock, None
        with torch.cuda.device(self.device):
            _lib.check(self._L.t1d_rollout_bb(self._ctx, C.byref(self._b), C.byref(p), int(n_steps),
                                              self.minutes_per_step, self.n_sub, self._stream()))
        if clock is not None:
            self._clock = clock + int(n_steps) * self.minutes_per_step
        self._hist_after_rollout()
        return bb_state

    def model_rhs(self, x, patient_idx, cho, insulin, last_qsto, last_food, math=1):
        """T1DPatient.model (t1dpatient.py:119-208) at m independent points on the device: x [13, m], the others [m]
        (cho grams eaten in the minute, insulin U/min as the model takes it, i.e. without the pump) -> dx/dt [13, m].
        math = 0: ocml tanh / IEEE divisions as the reference writes them; 1: the step kernels' arithmetic."""
        t = lambda v, dt=self.dtype: torch.as_tensor(v, dtype=dt).to(self.device).contiguous()
        x = t(x); m = x.shape[1]
        if x.shape[0] != 13:
            raise ValueError("x must have shape [13, m]")
        pid = t(patient_idx, torch.int32); args = [t(v) for v in (cho, insulin, last_qsto, last_food)]
        if pid.shape != (m,) or any(v.shape != (m,) for v in args):
            raise ValueError("patient_idx, cho, insulin, last_qsto, last_food must have m entries")
        if m and (int(pid.min()) < 0 or int(pid.max()) >= len(self.names)):
            raise ValueError("patient_idx out of range for the context's table of %d patients" % len(self.names))
        out = torch.empty(13, m, dtype=self.dtype, device=self.device)
        p = lambda v: C.c_void_p(v.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(self._L.t1d_model_rhs(self._ctx, self._b.dtype, m, int(math), p(x), p(pid), p(args[0]), p(args[1]), p(args[2]),
                                             p(args[3]), p(out), self._stream()))
        return out

    def philox_normals(self, n_draws, draw0=0, episode=1):
        """The normals the kernels draw in Philox mode -> float64 [n_draws, n] (for replay tests)."""
        out = torch.empty(n_draws, self.n, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.t1d_philox_normals(self._ctx, self.seed, self.env_offset, self.n, int(episode),
                                                  int(draw0), int(n_draws), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def sync(self, raise_on_status=True):
        """Wait for the stream; -> status bits (T1D_ST_*)."""
        st = C.c_int32(0)
        with torch.cuda.device(self.device):
            rc = self._L.t1d_sync(self._ctx, self._stream(), C.byref(st))
        if rc != 0 and (raise_on_status or st.value == 0):
            _lib.check(rc)
        return st.value

    # ------------------------------------------------------------------ checkpoint
    STATE_FORMAT = 4           # = the ABI version whose state words the checkpoint holds (prev_risk, cgm0: since 3; dbar: 4)

    def state_dict(self):
        sd = {k: getattr(self, k).clone() for k in _STATE_KEYS + ("cgm", "cgm0")}
        sd["format"] = self.STATE_FORMAT
        if self._hist is not None:
            sd.update({k: getattr(self, k).clone() for k in ("_hist", "_hist_pos", "_hist_cnt")})
        return sd

    def load_state_dict(self, sd):
        fmt = sd.get("format")
        if fmt != self.STATE_FORMAT or any(k not in sd for k in _STATE_KEYS + ("cgm", "cgm0")):
            raise _lib.T1DError("checkpoint format %r is not %d (checkpoints written before ABI 3 carry prev_cgm instead of "
                                "prev_risk and no cgm0, before ABI 4 no dbar): re-create it with this version" % (fmt, self.STATE_FORMAT))
        for k in _STATE_KEYS + ("cgm", "cgm0"):
            getattr(self, k).copy_(sd[k])
        if self._hist is not None:
            if all(k in sd for k in ("_hist", "_hist_pos", "_hist_cnt")):
                for k in ("_hist", "_hist_pos", "_hist_cnt"):
                    getattr(self, k).copy_(sd[k])
            else:
                # the checkpoint was taken without a CGM history: what the ring can honestly hold is the last observation
                self._hist_after_rollout()
        self._clock = None

    def close(self):
        if not self._closed and self._ctx:
            self._L.t1d_ctx_destroy(self._ctx)
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
